set -x
python -m pytest tests/test_gpu_parity_at_scale.py -x -q -m gpu 2>&1 | tail -25
python -m pytest tests/test_gpu_parity.py tests/test_scalar_operators.py -x -q -m gpu 2>&1 | tail -5
python scripts/asm_ab.py 60 2 asm_packed_codes 0 1
MFH_OPTIONS=asm_chunk_order=1 python scripts/asm_ab.py 60 2 asm_packed_codes 0 1
MFH_OPTIONS=matrix_storage=0 python scripts/asm_ab.py 60 2 asm_packed_codes 0 1
python scripts/asm_materials.py 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
PMC_UPPER_STORAGE=1 python $GRAFT_REPO_ROOT/scripts/pmc_collect.py 60 2>&1 | tail -12
cp $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic_n60_upper.json $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic_n60_upper_packed.json
PMC_UPPER_STORAGE=1 MFH_OPTIONS=asm_chunk_order=1 python $GRAFT_REPO_ROOT/scripts/pmc_collect.py 60 2>&1 | tail -12
cp $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic_n60_upper.json $GRAFT_REPO_ROOT/gpurun_out/pmc_traffic_n60_upper_packed_ordered.json
