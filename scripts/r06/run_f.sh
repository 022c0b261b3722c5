# round 6, GPU batch F: arena test, setup laps at configs[2], bench line with one_shot
O=gpurun_out/r06f
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_gpu_arena.py -x -q -m gpu > $O/tests_a.log 2>&1 < /dev/null
tail -3 $O/tests_a.log
timeout 600 python scripts/mg_setup_probe.py 60 2 > $O/mg_setup_60.log 2>&1 < /dev/null
timeout 900 python bench.py --no-strong-n1 --no-cpu --no-config3 > $O/bench_n1.json 2> $O/bench_n1.err < /dev/null
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06f/bench_n1.json").read().strip().splitlines()[-1])
print("kernel_ms", d["roofline"]["kernel_ms"], "value", d["value"]); print(json.dumps(d.get("one_shot"), indent=1))
PY
