# round 6, GPU batch B: the batched V-cycle (tests, configs[3] with and without it, kernel trace), hierarchy setup laps
O=gpurun_out/r06b
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_multigrid.py -x -q -m gpu > $O/tests_solver_mg.log 2>&1 < /dev/null
tail -5 $O/tests_solver_mg.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "homogenization or periodic" > $O/tests_hom.log 2>&1 < /dev/null
tail -3 $O/tests_hom.log
MFH_MG_TIMING=1 MFH_SOLVE_TIMING=1 timeout 600 python bench.py --leg config3 > $O/leg_config3_batch.json 2> $O/leg_config3_batch.err < /dev/null
MFH_OPTIONS="mg_batch=0" timeout 600 python bench.py --leg config3 > $O/leg_config3_seq.json 2> $O/leg_config3_seq.err < /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_config3 -- python $R/bench.py --leg config3 > $R/$O/leg_config3_prof.json 2> $R/$O/leg_config3_prof.err < /dev/null
f=$(find $R/$O/prof_config3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/config3_kernel_stats.csv
f=$(find $R/$O/prof_config3 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $R/scripts/trace_by_level.py "$f" > $R/$O/config3_trace_by_level.txt 2>&1
rm -rf $R/$O/prof_config3
cd $R
python - <<'PY'
import json
for n in ("batch","seq"):
    try:
        d=json.load(open("gpurun_out/r06b/leg_config3_%s.json"%n))
        print(n, d["wall_s"]["cell_problems"], d["cell_problems"]["iterations"], [round(x,1) for x in d["cell_problems"]["solve_ms"]], d["cell_problems"]["hierarchy_setup_ms"], d["Ch_diag"][:3])
    except Exception as e:
        print(n, "failed", e)
PY
