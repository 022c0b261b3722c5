#!/bin/bash
# do the vector kernels care where the vectors start relative to each other? arena granularity (default 2 MiB: every large buffer starts on a 2 MiB boundary)
mkdir -p gpurun_out/r06af
for rep in 1 2 3; do
for g in 0 4 64 260; do
  if [ $g = 0 ]; then echo "granularity default (2 MiB)"; timeout 300 python scripts/r06/grid_cap_probe.py 2>&1 | grep "^cap"
  else echo "MFH_ARENA_GRAN_KB=$g"; MFH_ARENA_GRAN_KB=$g timeout 300 python scripts/r06/grid_cap_probe.py 2>&1 | grep "^cap"; fi
done; done | tee gpurun_out/r06af/gran.txt
