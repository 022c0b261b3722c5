set -x
MFH_MG_TIMING=1 MFH_TL_TIMING=1 python scripts/mg_probe.py 60 1,3,0.3,0.1,1 2>&1 | cut -c1-170
