#!/bin/bash
mkdir -p gpurun_out/r06ac
for rep in 1 2; do
for g in 2048 1024 1280 1536 1792 2304 2560 3072 4096 6144 8192; do
  MFH_GRID_CAP=$g timeout 300 python scripts/r06/grid_cap_probe.py 2>&1 | grep "^cap"
done; done | tee gpurun_out/r06ac/grid_cap.txt
