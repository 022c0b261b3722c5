#!/bin/bash
# host-side laps of the configs[3] cell problems (MFH_SOLVE_TIMING=1)
mkdir -p gpurun_out/r06v
MFH_SOLVE_TIMING=1 timeout 600 python bench.py --leg config3 > gpurun_out/r06v/config3.json 2> gpurun_out/r06v/config3.err
tail -80 gpurun_out/r06v/config3.err
