set -x
python -m pytest tests/test_gpu_parity_at_scale.py -x -q -m gpu 2>&1 | tail -4
MFH_OPTIONS=asm_chunk_order=1 python scripts/asm_ab.py 60 2 xcd_swizzle 0 1
MFH_OPTIONS=asm_chunk_order=0 python scripts/asm_ab.py 60 2 xcd_swizzle 0 1
for v in u4 u1 nosb8; do MESHFEM_HIP_LIB=meshfem_amd/variants/libmeshfem_hip_$v.so python scripts/asm_ab.py 60 2 asm_chunk_order 0 1; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z_0-9a-z]*\|TA_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*\|TD_[A-Z_0-9a-z]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt
PMC_UPPER_STORAGE=1 MFH_OPTIONS=asm_chunk_order=1,xcd_swizzle=1 python $GRAFT_REPO_ROOT/scripts/pmc_collect.py 60 2>&1 | head -8
