# Round-6 record (run on the GPU box from the repo root). One rocprofv3 --kernel-trace --stats per WORKLOAD (as in round 5): configs[1],
# configs[2], configs[3], configs[4] in one context -- each a `bench.py --leg` process of its own, nothing else in the trace --, the PMC
# traffic of the round-6 kernels (one counter per run), the default bench line, forced multi-rank lines on one GPU (preflight + both solve legs).
# Every step has its own timeout and reads no stdin.
O=gpurun_out/final_r06
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for leg in config2 config1 config3 strong_n1; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$leg -- python $R/bench.py --leg $leg > $R/$O/leg_$leg.json 2> $R/$O/leg_$leg.err < /dev/null
  f=$(find $R/$O/prof_$leg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/r06_${leg}_kernel_stats.csv
  f=$(find $R/$O/prof_$leg -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/scripts/trace_by_level.py "$f" > $R/$O/r06_${leg}_kernel_trace_summary.txt 2>&1
  rm -rf $R/$O/prof_$leg
done
PMC_UPPER_STORAGE=1 timeout 900 python $R/scripts/pmc_collect.py 60 > $R/$O/pmc_upper.out 2>&1 < /dev/null
cp $R/gpurun_out/pmc_traffic_n60_upper.json $R/$O/r06_pmc_traffic_n60_upper_storage.json
cp $R/gpurun_out/pmc/FETCH_SIZE_n60_counter_collection.csv $R/$O/r06_pmc_FETCH_SIZE_n60_upper.csv; cp $R/gpurun_out/pmc/WRITE_SIZE_n60_counter_collection.csv $R/$O/r06_pmc_WRITE_SIZE_n60_upper.csv
timeout 900 python $R/scripts/pmc_collect.py 60 > $R/$O/pmc_full.out 2>&1 < /dev/null
cp $R/gpurun_out/pmc_traffic_n60.json $R/$O/r06_pmc_traffic_n60.json
cd $R
timeout 1200 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err < /dev/null
timeout 900 python bench.py --gpus 2 --ranks-per-gpu-ok --scaling strong --grid 48 --no-cpu --no-weak > $O/bench_n2_strong48.out 2> $O/bench_n2_strong48.err < /dev/null
timeout 1500 python bench.py --gpus 8 --ranks-per-gpu-ok --no-cpu --no-weak > $O/bench_n8_forced.out 2> $O/bench_n8_forced.err < /dev/null
ls -la $O | head -40
