O=gpurun_out/r06l
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
MFH_POOL_TRACE=1 timeout 1200 python bench.py --no-cpu > $O/bench_n1.json 2> $O/bench_n1.err < /dev/null
grep "arena" $O/bench_n1.err | tail -60
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06l/bench_n1.json").read().strip().splitlines()[-1])
s=d["strong_scaling_n1"]
print("fresh", {k:round(v,3) for k,v in s["setup"].items() if isinstance(v,float)})
print("warm", {k:round(v,3) for k,v in s["warm_process"]["setup"].items() if isinstance(v,float)})
print("kernel", d["roofline"]["kernel_ms"], s["kernel_ms"], s["warm_process"]["kernel_ms"])
PY
