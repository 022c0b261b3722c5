#!/bin/bash
mkdir -p gpurun_out/r06ao
for rep in 1 2 3; do
  timeout 300 python bench.py --leg config2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench --leg config2 (reservation first)      kernel %.4f ms' % d['kernel_ms'])"
  MFH_BENCH_NO_RESERVE=1 timeout 300 python bench.py --leg config2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench --leg config2 MFH_BENCH_NO_RESERVE=1   kernel %.4f ms' % d['kernel_ms'])"
  for m in reserve_torch plain; do timeout 300 python scripts/r06/placement_order_probe.py $m 2>&1 | grep kernel; done
done | tee gpurun_out/r06ao/order2.txt
