O=gpurun_out/r06h
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 1500 python scripts/r06/auto_precond_probe.py 24 > $O/auto_precond_24.jsonl 2> $O/auto_precond_24.err < /dev/null
cat $O/auto_precond_24.jsonl
timeout 600 python -m pytest tests/test_gpu_multigrid.py tests/test_cpp_simulate_cli.py tests/test_cli_io.py -x -q -m gpu > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
