# round 6, GPU batch G: device-side neumann load (parity tests), dense-level size probe, arena test
O=gpurun_out/r06g
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_gpu_arena.py tests/test_gpu_parity.py tests/test_constrained_solve.py tests/test_cpp_simulate_cli.py -x -q -m gpu > $O/tests_a.log 2>&1 < /dev/null
tail -3 $O/tests_a.log
timeout 600 python scripts/mg_dense_probe.py 60 1200 600 300 150 > $O/mg_dense_60.log 2>&1 < /dev/null
cat $O/mg_dense_60.log | grep mg_dense
MFH_SOLVE_TIMING=1 timeout 600 python scripts/mg_setup_probe.py 60 1 2>&1 | grep "mfh solve\|rep" > $O/solve_laps_60.log
cat $O/solve_laps_60.log
