O=gpurun_out/r06k
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
MFH_POOL_TRACE=1 MFH_SYM_TIMING=1 timeout 900 python bench.py --leg strong_n1 --no-solve > $O/strong_trace.json 2> $O/strong_trace.err < /dev/null
grep -v "^\[symbolic\]  " $O/strong_trace.err | tail -40
