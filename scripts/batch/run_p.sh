set -x
for a in 1.2 1.3 1.5 1.6; do MFH_OPTIONS=mg_over_correction=$a python scripts/mg_probe.py 60 1,1,0.3,0.3,1 1,1,0.3,0.3,1,1,0.3 1,1,0.3,0.3,1,3,0.1 2>&1 | grep multigrid | cut -c1-150; done
MFH_OPTIONS=mg_over_correction=1.4 python scripts/mg_probe.py 60 1,1,0.3,0.3,2 1,1,0.5,0.5,1 1,1,0.15,0.15,1 2>&1 | grep multigrid | cut -c1-150
cd /tmp && export TMPDIR=/tmp
MFH_OPTIONS=mg_over_correction=1.4,mg_steps_coarse=1,mg_ratio_coarse=0.3 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mgprof -- python $GRAFT_REPO_ROOT/scripts/mg_profile.py > $GRAFT_REPO_ROOT/gpurun_out/mgprof.out 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/mgprof.out | cut -c1-300
f=$(find $GRAFT_REPO_ROOT/gpurun_out/mgprof -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-200
