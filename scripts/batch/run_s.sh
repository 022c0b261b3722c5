set -x
timeout 400 python -m pytest tests/test_gpu_multigrid.py -x -q -m gpu 2>&1 | tail -6
MFH_MG_TIMING=1 timeout 300 python scripts/mg_probe.py 60 1,1,0.3,0.3,1 2>&1 | grep "multigrid\|two-level" | cut -c1-250
