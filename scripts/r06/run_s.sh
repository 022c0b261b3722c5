O=gpurun_out/r06s
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- python $R/bench.py --leg config2 > $R/$O/leg.json 2> $R/$O/leg.err < /dev/null
f=$(find $R/$O/prof -name "*kernel_trace.csv" | head -1)
python $R/scripts/r06/copy_neighbours.py "$f"
rm -rf $R/$O/prof
