#!/bin/bash
# restriction / prolongation kernels with their loads issued together: tests, per-iteration time, kernel durations
mkdir -p gpurun_out/r06aa
timeout 900 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_solver.py -x -q 2>&1 | tail -3 | tee gpurun_out/r06aa/tests.log
timeout 300 python scripts/r06/fuse_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06aa/config2.txt
for rep in 1 2; do
  timeout 300 python bench.py --leg config3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cell_problems']
print('configs[3]: cell_problems %.3f s, device time of the batch %.1f ms, iterations %s' % (d['wall_s']['cell_problems'], c['device_ms_all_solves'], c['iterations']))" | tee -a gpurun_out/r06aa/config3.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06aa/trace -- python $GRAFT_REPO_ROOT/bench.py --leg config3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r06aa/trace/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print("%-70s calls %6s avg %9.1f us" % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
