# round 6, GPU batch D: tile-geometry probe, arena values class (tests + bench), rest of the suite after the failed test
O=gpurun_out/r06d
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_arena.py -x -q -m gpu > $O/tests_a.log 2>&1 < /dev/null
tail -3 $O/tests_a.log
timeout 1500 python scripts/r06/tile_probe.py > $O/tile_probe.txt 2>&1 < /dev/null
cat $O/tile_probe.txt
timeout 900 python bench.py --no-strong-n1 --no-cpu > $O/bench_n1.json 2> $O/bench_n1.err < /dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06d/bench_n1.json").read().strip().splitlines()[-1])
print("kernel_ms", d["roofline"]["kernel_ms"], "value", d["value"], "config3", d["variants"]["config3_homogenization"]["wall_s"])
PY
timeout 1500 python -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_solver.py --deselect tests/test_gpu_arena.py -k "not parity and not multigrid" > $O/suite_rest.log 2>&1 < /dev/null
tail -3 $O/suite_rest.log
