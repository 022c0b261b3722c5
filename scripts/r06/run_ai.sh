#!/bin/bash
# spread of the assembly kernel's profile average over processes: five rocprofv3 --kernel-trace --stats runs of `bench.py --leg config2`; the MEDIAN run's files are the committed ones
O=gpurun_out/r06ai; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p$i -- python $R/bench.py --leg config2 > $R/$O/leg_$i.json 2> /dev/null < /dev/null
  f=$(find $R/$O/p$i -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/stats_$i.csv
  f=$(find $R/$O/p$i -name "*kernel_trace.csv" | head -1); python $R/scripts/trace_by_level.py "$f" > $R/$O/summary_$i.txt 2>&1
  rm -rf $R/$O/p$i
  python - <<PY
import csv, json
rows = list(csv.DictReader(open("$R/$O/stats_$i.csv")))
r = [x for x in rows if "k_assemble_gather<3, 2, 0, true, false>" in x["Name"]][0]
d = json.loads(open("$R/$O/leg_$i.json").read().strip().splitlines()[-1])
print("run $i: k_assemble_gather profile avg %.4f ms over %s launches; the leg's own HIP-event time %.4f ms; multigrid solve %.1f ms" % (float(r["AverageNs"]) / 1e6, r["Calls"], d["kernel_ms"], d["pcg_multigrid"]["solve_ms"]))
PY
done | tee $R/$O/spread.txt
