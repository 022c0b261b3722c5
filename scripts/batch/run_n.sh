set -x
MFH_MG_TIMING=1 python scripts/mg_probe.py 12 1,3,0.3,0.1,1 2>&1 | cut -c1-330
MFH_MG_TIMING=1 python scripts/mg_probe.py 60 1,3,0.3,0.1,1 1,2,0.3,0.15,1 1,1,0.3,0.3,1 1,2,0.3,0.15,1,3,0.1 1,2,0.3,0.15,1,1,0.3 2,2,0.3,0.15,1 1,2,0.3,0.15,2 2>&1 | cut -c1-330
MFH_OPTIONS=mg_agg_target=0 python scripts/mg_probe.py 60 1,3,0.3,0.1,1 2>&1 | cut -c1-230
python -m pytest tests/test_gpu_multigrid.py -x -q -m gpu 2>&1 | tail -5
