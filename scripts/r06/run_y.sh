#!/bin/bash
mkdir -p gpurun_out/r06y
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06y/trace -- python $GRAFT_REPO_ROOT/scripts/r06/spmv_p1_probe.py 60 2>&1 | grep -v "amdgpu.ids\|rocprofv3" | tail -8
