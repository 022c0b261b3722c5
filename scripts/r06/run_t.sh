O=gpurun_out/r06t
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python scripts/mg_setup_probe.py 119 1 > $O/mg_setup_119.log 2>&1 < /dev/null
grep -v "^\[symbolic\]   \|operator lists\]\|element order\]" $O/mg_setup_119.log | tail -60
