#!/bin/bash
mkdir -p gpurun_out/r06ae
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14; do
  MFH_PRINT_VECS=1 timeout 300 python scripts/r06/grid_cap_probe.py 2>&1 | grep "^cap\|vecs" | sort -u | head -4
  echo
done | tee gpurun_out/r06ae/vecs.txt
