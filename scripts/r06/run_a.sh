# round 6, GPU batch A: orthotropic operator at 3 waves (parity + time), where configs[3]'s wall time goes, a baseline bench line
O=gpurun_out/r06a
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ortho or anisotropic or homogenization or heterogeneous or shape or strain" > $O/parity_subset.log 2>&1 < /dev/null
tail -3 $O/parity_subset.log
timeout 600 python scripts/hom_profile.py 44 > $O/hom_profile.log 2>&1 < /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_config3 -- python $R/bench.py --leg config3 > $R/$O/leg_config3.json 2> $R/$O/leg_config3.err < /dev/null
f=$(find $R/$O/prof_config3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/config3_kernel_stats.csv
rm -rf $R/$O/prof_config3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_batch -- python $R/scripts/r06/batch_probe.py 44 30 > $R/$O/batch_probe.log 2>&1 < /dev/null
f=$(find $R/$O/prof_batch -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/batch_probe_kernel_stats.csv
rm -rf $R/$O/prof_batch
cd $R
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err < /dev/null
tail -c 600 $O/bench_n1.json
