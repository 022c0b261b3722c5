O=gpurun_out/r06r
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
tail -2 $O/smoke.log
SECONDS=0; python bench.py > $O/bench_n1.json 2> $O/bench_n1.err < /dev/null
echo "bench wall ${SECONDS} s"
tail -c 400 $O/bench_n1.json
