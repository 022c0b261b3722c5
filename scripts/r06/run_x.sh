#!/bin/bash
# kernel timeline of multigrid solves at configs[2] (per-call durations and gaps inside the graph replays)
mkdir -p gpurun_out/r06x
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06x/trace -- python $GRAFT_REPO_ROOT/scripts/r06/additive_probe.py 60 2>&1 | grep -v amdgpu.ids | tail -5
cd $GRAFT_REPO_ROOT
find gpurun_out/r06x -name "*kernel_trace.csv" | head
