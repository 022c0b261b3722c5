set -x
for a in 1.0 1.4 1.8 2.2; do MFH_OPTIONS=mg_over_correction=$a python scripts/mg_probe.py 60 1,1,0.3,0.3,1 1,2,0.3,0.15,1 2>&1 | grep multigrid | cut -c1-150; done
for t in 8 64; do MFH_OPTIONS=mg_agg_target=$t,mg_over_correction=1.4 python scripts/mg_probe.py 60 1,1,0.3,0.3,1 1,2,0.3,0.15,1 2>&1 | grep multigrid | cut -c1-150; done
MFH_OPTIONS=mg_agg_target=0,mg_over_correction=1.5 python scripts/mg_probe.py 60 1,3,0.3,0.1,1 2>&1 | grep multigrid | cut -c1-150
