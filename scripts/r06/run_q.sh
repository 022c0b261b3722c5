O=gpurun_out/r06q
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/suite.log 2>&1 < /dev/null
tail -4 $O/suite.log
