#!/bin/bash
mkdir -p gpurun_out/r06an
for rep in 1 2 3; do
for m in plain reserve torch_reserve reserve_torch torch_plain; do
  timeout 300 python scripts/r06/placement_order_probe.py $m 2>&1 | grep kernel
done; done | tee gpurun_out/r06an/order.txt
