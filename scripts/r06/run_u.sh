O=gpurun_out/r06u
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python bench.py --leg strong_n1 > $O/strong.json 2> $O/strong.err < /dev/null
python - <<'PY'
import json
s=json.load(open("gpurun_out/r06u/strong.json"))
print({k:round(v,3) for k,v in s["setup"].items() if isinstance(v,float)}); print(s["one_shot"]); print(s["pcg_multigrid"])
PY
timeout 600 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_arena.py -x -q -m gpu > $O/tests.log 2>&1 < /dev/null
tail -2 $O/tests.log
