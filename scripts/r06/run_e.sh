# round 6, GPU batch E: arena tests, tile-geometry probe
O=gpurun_out/r06e
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_gpu_arena.py tests/test_gpu_solver.py -x -q -m gpu > $O/tests_a.log 2>&1 < /dev/null
tail -3 $O/tests_a.log
timeout 1500 python scripts/r06/tile_probe.py > $O/tile_probe.txt 2>&1 < /dev/null
cat $O/tile_probe.txt
