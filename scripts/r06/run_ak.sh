#!/bin/bash
# option mg_dinv_fp32: tests, then on / off at configs[2]'s shape and on the configs[3] cell problems
mkdir -p gpurun_out/r06ak
timeout 900 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_solver.py tests/test_gpu_deterministic.py -x -q 2>&1 | grep -v "version\|Hostname\|Librccl\|^$" | tail -4 | tee gpurun_out/r06ak/tests.log
timeout 300 python scripts/r06/fuse_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06ak/config2.txt
for rep in 1 2; do
for f in 1 0; do
  MFH_OPTIONS="mg_dinv_fp32=$f" timeout 300 python bench.py --leg config3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cell_problems']
print('configs[3] mg_dinv_fp32 $f: cell_problems %.3f s, device time of the batch %.1f ms, iterations %s, Ch[0] %.12g' % (d['wall_s']['cell_problems'], c['device_ms_all_solves'], c['iterations'], d['Ch_diag'][0]))" | tee -a gpurun_out/r06ak/config3.txt
done; done
