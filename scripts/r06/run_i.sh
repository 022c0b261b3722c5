O=gpurun_out/r06i
mkdir -p $O
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 600 python -m pytest tests/test_gpu_multigrid.py -x -q -m gpu -k "stretched" > $O/tests.log 2>&1 < /dev/null
tail -3 $O/tests.log
timeout 2400 python scripts/r06/asm_ablation.py > $O/asm_ablation.txt 2>&1 < /dev/null
cat $O/asm_ablation.txt
