cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
grep -c . $R/gpurun_out/counters_list.txt
MFH_BENCH_FORCE_DISTRIBUTED=1 timeout 600 python $R/bench.py --steps 5 --warmup 1 --grid 30 > $R/gpurun_out/bench_forced_dist.json 2> $R/gpurun_out/bench_forced_dist.err; echo rc=$?; tail -c 1500 $R/gpurun_out/bench_forced_dist.json
