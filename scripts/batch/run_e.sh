set -x
MESHFEM_HIP_LIB=meshfem_amd/variants/libmeshfem_hip_skipcut.so python scripts/asm_ab.py 60 2 asm_chunk_order 0 1
python scripts/asm_ab.py 60 2 asm_chunk_order 0 1
MESHFEM_HIP_LIB=meshfem_amd/variants/libmeshfem_hip_skipcut.so MFH_OPTIONS=matrix_storage=0 python scripts/asm_ab.py 60 2 asm_chunk_order 0
MFH_OPTIONS=matrix_storage=0 python scripts/asm_ab.py 60 2 asm_chunk_order 0
