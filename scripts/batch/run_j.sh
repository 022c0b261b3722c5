set -x
python -m pytest tests/test_gpu_multigrid.py -x -q -m gpu 2>&1 | tail -30
MFH_MG_TIMING=1 python scripts/mg_probe.py 60 1,3,0.3,0.1,1 2>&1 | cut -c1-170
