cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3
timeout 1200 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err
echo "bench rc=$?"; cat gpurun_out/bench_r01.json; tail -5 gpurun_out/bench_r01.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/bench_r01_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r01.err
echo "prof rc=$?"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_r01.err
find $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -name "*stats*" | head; 
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -name "*kernel_stats.csv" | head -1); head -20 "$f"
# keep only the small summaries (the raw trace can be large)
find $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -name "*kernel_trace.csv" -size +20M -delete
