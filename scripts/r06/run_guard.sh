#!/bin/bash
# the -m gpu suite with guard bytes behind every device buffer (MFH_ARENA_GUARD=1, mfh_pool.cpp): a kernel writing past the end of its buffer aborts the run
mkdir -p gpurun_out/r06guard
MFH_ARENA_GUARD=1 timeout 2700 python -m pytest tests/ -q -m gpu --ignore=tests/test_gpu_arena.py --ignore=tests/test_gpu_threads.py > gpurun_out/r06guard/suite.log 2>&1 < /dev/null
grep -v "version\|Hostname\|Librccl\|^$" gpurun_out/r06guard/suite.log | tail -8
grep -c "arena guard" gpurun_out/r06guard/suite.log
