#!/bin/bash
mkdir -p gpurun_out/r06ah
timeout 1500 python -m pytest tests/test_gpu_solver.py tests/test_gpu_multigrid.py tests/test_gpu_deterministic.py tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_distributed_multigrid.py -x -q 2>&1 | grep -v "version\|Hostname\|Librccl\|^$" | tail -6 | tee gpurun_out/r06ah/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06ah/p -- python $GRAFT_REPO_ROOT/scripts/r06/grid_cap_probe.py 2>&1 | grep "^cap"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r06ah/p/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-80s calls %6s avg %9.1f us" % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3))
PY
