#!/bin/bash
# does the profiler change the assembly kernel's time? `bench.py --leg config2` with and without rocprofv3 --kernel-trace --stats, alternating, one box
O=gpurun_out/r06am; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 600 python $R/bench.py --leg config2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i without the profiler: HIP-event kernel time %.4f ms, step %.4f ms' % (d['kernel_ms'], d['ms_per_step']))"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/p$i -- python $R/bench.py --leg config2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i under rocprofv3 --kernel-trace --stats: HIP-event kernel time %.4f ms, step %.4f ms' % (d['kernel_ms'], d['ms_per_step']))"
  rm -rf $R/$O/p$i
done | tee $R/$O/with_without.txt
