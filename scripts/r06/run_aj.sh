#!/bin/bash
# the fuzzers on the final library (new seed ranges): unstructured meshes and free bodies against the oracle, scatter variants
mkdir -p gpurun_out/r06aj
timeout 1500 python scripts/fuzz_unstructured.py 2000 150 > gpurun_out/r06aj/unstructured.log 2>&1; tail -2 gpurun_out/r06aj/unstructured.log
timeout 900 python scripts/fuzz_free_body.py 2000 40 > gpurun_out/r06aj/free_body.log 2>&1; tail -2 gpurun_out/r06aj/free_body.log
timeout 900 python scripts/fuzz_scatter.py 2000 30 > gpurun_out/r06aj/scatter.log 2>&1; tail -2 gpurun_out/r06aj/scatter.log
