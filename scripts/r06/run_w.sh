#!/bin/bash
mkdir -p gpurun_out/r06w
{
timeout 300 python scripts/r06/additive_probe.py 60
for w in 0.7 1.0 1.4 2.0; do MFH_MG_ADDITIVE=$w timeout 300 python scripts/r06/additive_probe.py 60; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06w/additive.txt
