set -x
python scripts/mg_probe.py 12 2,4,0.25,0.05
python scripts/mg_probe.py 60 2,4,0.25,0.05 1,4,0.3,0.05 2,6,0.25,0.03 2,3,0.25,0.1 3,4,0.15,0.05 2,8,0.25,0.02
