#!/bin/bash
# option mg_fuse: the test, then on / off at configs[2]'s shape and at the configs[3] cell problems
mkdir -p gpurun_out/r06z
timeout 900 python -m pytest tests/test_gpu_multigrid.py -x -q -k "fused or oracle_direct or nonzero_dirichlet" 2>&1 | tail -5 | tee gpurun_out/r06z/tests.log
timeout 300 python scripts/r06/fuse_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06z/fuse_config2.txt
for rep in 1 2; do
for f in 1 0; do
  MFH_OPTIONS="mg_fuse=$f" timeout 300 python bench.py --leg config3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cell_problems']
print('configs[3] mg_fuse $f: cell_problems %.3f s, device time of the batch %.1f ms, iterations %s' % (d['wall_s']['cell_problems'], c['device_ms_all_solves'], c['iterations']))" | tee -a gpurun_out/r06z/fuse_config3.txt
done; done
