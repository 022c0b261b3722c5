# the whole -m gpu suite (as the driver runs it) + the default bench line
O=gpurun_out/r06suite
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/suite.log 2>&1 < /dev/null
tail -5 $O/suite.log
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err < /dev/null
tail -c 300 $O/bench_n1.json
