# End-of-round record (run on the GPU box from the repo root): default bench, bench under rocprofv3, PMC traffic of the assembly kernel,
# kernel stats of a multigrid solve, forced 2-rank weak / strong runs on one GPU, the N = 1 point of the strong-scaling curve (configs[4]).
# Every step has its own timeout and reads no stdin.
mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err < /dev/null
timeout 600 python bench.py --gpus 2 --ranks-per-gpu-ok --no-cpu > gpurun_out/final/bench_n2_weak.out 2> gpurun_out/final/bench_n2_weak.err < /dev/null
timeout 600 python bench.py --gpus 2 --ranks-per-gpu-ok --scaling strong --grid 48 --no-cpu > gpurun_out/final/bench_n2_strong48.out 2> gpurun_out/final/bench_n2_strong48.err < /dev/null
timeout 600 python bench.py --gpus 4 --ranks-per-gpu-ok --scaling strong --grid 48 --no-cpu > gpurun_out/final/bench_n4_strong48.out 2> gpurun_out/final/bench_n4_strong48.err < /dev/null
timeout 600 python bench.py --gpus 1 --scaling strong --grid 48 --no-cpu > gpurun_out/final/bench_n1_strong48.out 2> gpurun_out/final/bench_n1_strong48.err < /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -- python $R/bench.py --no-cpu > $R/gpurun_out/final/bench_prof.json 2> $R/gpurun_out/final/bench_prof.err < /dev/null
f=$(find $R/gpurun_out/final/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/final/r03_bench_n1_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/mgprof -- python $R/scripts/mg_profile.py > $R/gpurun_out/final/mgprof.out 2>&1 < /dev/null
f=$(find $R/gpurun_out/final/mgprof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/final/r03_multigrid_solve_kernel_stats.csv
PMC_UPPER_STORAGE=1 timeout 900 python $R/scripts/pmc_collect.py 60 > $R/gpurun_out/final/pmc_upper.out 2>&1 < /dev/null
cp $R/gpurun_out/pmc_traffic_n60_upper.json $R/gpurun_out/final/r03_pmc_traffic_n60_upper_storage.json
PMC_UPPER_STORAGE=1 MFH_OPTIONS=asm_chunk_order=1,xcd_swizzle=1 timeout 900 python $R/scripts/pmc_collect.py 60 > $R/gpurun_out/final/pmc_upper_ordered.out 2>&1 < /dev/null
cp $R/gpurun_out/pmc_traffic_n60_upper.json $R/gpurun_out/final/r03_pmc_traffic_n60_upper_storage_chunk_order_xcd.json
cd $R
timeout 1500 python bench.py --gpus 1 --scaling strong --no-cpu > gpurun_out/final/bench_n1_strong119.out 2> gpurun_out/final/bench_n1_strong119.err < /dev/null
ls -la gpurun_out/final | head -40
