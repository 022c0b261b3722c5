mkdir -p gpurun_out/final
timeout 600 python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err
timeout 600 python bench.py --gpus 2 --ranks-per-gpu-ok > gpurun_out/final/bench_n2.out 2> gpurun_out/final/bench_n2.err
tail -c 600 gpurun_out/final/bench_n2.out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/final/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/final/bench_prof.err
ls -R $GRAFT_REPO_ROOT/gpurun_out/final/prof | head
