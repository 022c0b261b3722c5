#!/bin/bash
# interface rows' share of x.(Kx) taken in k_mf_cluster: tests of every solver path, then per-iteration times
mkdir -p gpurun_out/r06ag
timeout 1500 python -m pytest tests/test_gpu_solver.py tests/test_gpu_multigrid.py tests/test_gpu_deterministic.py tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_gpu_distributed_multigrid.py -x -q 2>&1 | tail -4 | tee gpurun_out/r06ag/tests.log
timeout 300 python scripts/r06/grid_cap_probe.py 2>&1 | grep "^cap" | tee gpurun_out/r06ag/iter.txt
timeout 300 python bench.py --leg config1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config1 pcg', d['pcg_block_jacobi']['iterations'], d['pcg_block_jacobi']['ms_per_iteration'], d['operator'])" | tee -a gpurun_out/r06ag/iter.txt
