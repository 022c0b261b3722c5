#!/bin/bash
mkdir -p gpurun_out/r06ab
for g in 2048 1024 1536 3072 4096 8192 16384; do
  echo "MFH_FUSED_GRID=$g"; MFH_FUSED_GRID=$g timeout 300 python scripts/r06/fuse_probe.py 2>&1 | grep "mg_fuse 1" | tail -2
done | tee gpurun_out/r06ab/grid.txt
