#!/bin/bash
# which kernels differ between the fast and the slow state of a process? six processes under rocprofv3 --stats
mkdir -p gpurun_out/r06ad
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06ad/p$i -- python $GRAFT_REPO_ROOT/scripts/r06/grid_cap_probe.py 2>&1 | grep "^cap"
done | tee $GRAFT_REPO_ROOT/gpurun_out/r06ad/summary.txt
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
names = ["k_mf_cluster<3, 2, 0, 1", "k_mf_cluster<3, 2, 0, 0", "k_mf_rows<3, 1", "k_mf_rows<3, 0", "k_pcg_update<3, false, false", "k_pcg_update<3, false, true", "k_pcg_direction", "k_mg_cheb_rz", "k_mg_restrict", "k_mg_prolong_add<", "k_spmv<3, 0", "k_st_spmv"]
for i in range(1, 9):
    f = glob.glob('gpurun_out/r06ad/p%d/**/*kernel_stats.csv' % i, recursive=True)
    if not f: continue
    rows = list(csv.DictReader(open(f[0])))
    out = []
    for n in names:
        r = [x for x in rows if n in x['Name']]
        out.append("%7.1f" % (sum(float(x['AverageNs']) * int(x['Calls']) for x in r) / max(1, sum(int(x['Calls']) for x in r)) / 1e3))
    print("p%d " % i + " ".join(out))
print("    " + " ".join("%7s" % n[-7:] for n in names))
PY
