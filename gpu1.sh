cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 2>&1 | tail -40
timeout 300 python scripts/perf_probe.py 30 2 2>&1 | tail -8
