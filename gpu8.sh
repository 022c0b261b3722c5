cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu --timeout 300 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 2>&1 | tail -2
timeout 1200 python bench.py > gpurun_out/bench_r01b.json 2> gpurun_out/bench_r01b.err; echo "bench rc=$?"; cat gpurun_out/bench_r01b.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01b -o bench -- python $R/bench.py --no-cpu > $R/gpurun_out/bench_r01b_prof.json 2> $R/gpurun_out/prof_r01b.err
echo "prof rc=$?"; head -8 $R/gpurun_out/prof_r01b/bench_kernel_stats.csv
rm -f $R/gpurun_out/prof_r01b/bench_kernel_trace.csv
