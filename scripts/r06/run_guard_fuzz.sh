#!/bin/bash
mkdir -p gpurun_out/r06guard
export MFH_ARENA_GUARD=1
timeout 1200 python scripts/fuzz_unstructured.py 3000 60 > gpurun_out/r06guard/unstructured.log 2>&1; tail -1 gpurun_out/r06guard/unstructured.log
timeout 900 python scripts/fuzz_free_body.py 3000 20 > gpurun_out/r06guard/free_body.log 2>&1; tail -1 gpurun_out/r06guard/free_body.log
timeout 900 python scripts/fuzz_scatter.py 3000 12 > gpurun_out/r06guard/scatter.log 2>&1; tail -1 gpurun_out/r06guard/scatter.log
grep -c "arena guard" gpurun_out/r06guard/*.log
