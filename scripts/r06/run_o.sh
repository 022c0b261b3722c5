O=gpurun_out/r06o
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python scripts/r06/rot_probe.py > $O/rot_probe.txt 2>&1 < /dev/null
cat $O/rot_probe.txt
