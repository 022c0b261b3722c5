set -x
timeout 300 python -m pytest tests/test_gpu_multigrid.py -x -q -m gpu 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mgprof -- python $GRAFT_REPO_ROOT/scripts/mg_profile.py > $GRAFT_REPO_ROOT/gpurun_out/mgprof.out 2>&1 < /dev/null
grep "iterations" $GRAFT_REPO_ROOT/gpurun_out/mgprof.out | cut -c1-300
f=$(find $GRAFT_REPO_ROOT/gpurun_out/mgprof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r03_multigrid_solve_kernel_stats.csv; head -28 "$f" | cut -c1-180; fi
