set -x
python -m pytest tests/test_gpu_parity.py tests/test_host_logic.py tests/test_gpu_multigrid.py -x -q -m gpu 2>&1 | tail -3
python scripts/ds_probe.py 10 2 0 1 2>&1 | grep -v "^$" | head -30
python scripts/ds_probe.py 10 2 16
python scripts/ds_probe.py 10 2 8
MFH_MESH_TIMING=1 python bench.py --no-orderings --no-cpu > gpurun_out/r03_bench_b.json 2> gpurun_out/r03_bench_b.err; tail -20 gpurun_out/r03_bench_b.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_bench_b.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["setup"], d["roofline"]["kernel_ms"])
for k in ("pcg","pcg_two_level","pcg_multigrid"):
    print(k, {a:b for a,b in d[k].items() if a in ("iterations","solve_ms","ms_per_iteration","hierarchy_setup_ms","coarse_setup_ms")})
PY
