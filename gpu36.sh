cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 200 gpurun_out/bench_final.json; echo
cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-orderings --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/rocprof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*kernel_trace.csv" -delete
timeout 1500 python scripts/config4_homogenization.py 44 --skip-bj 2>&1 | tail -2
MFH_BENCH_FORCE_DISTRIBUTED=1 timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['pcg']))"
