set -x
python -m pytest tests/test_gpu_parity_at_scale.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_free or cluster or heterogeneous or anisotropic" 2>&1 | tail -5
python -m pytest tests/test_gpu_distributed.py tests/test_gpu_solver.py -x -q -m gpu 2>&1 | tail -5
python scripts/op_opts.py 60 mf_quad_lanes 0 1 0 1
MESHFEM_HIP_LIB=meshfem_amd/variants/libmeshfem_hip_q5.so python scripts/op_opts.py 60 mf_quad_lanes 1 0 1
MESHFEM_HIP_LIB=meshfem_amd/variants/libmeshfem_hip_q6.so python scripts/op_opts.py 60 mf_quad_lanes 1 0 1
MFH_OPTIONS=mf_geometry_from_vertices=0 python scripts/op_opts.py 60 mf_quad_lanes 0 1
python scripts/asm_ab.py 60 2 asm_chunk_order 0 1
MFH_OPTIONS=xcd_swizzle=32 python scripts/asm_ab.py 60 2 asm_chunk_order 0 1
MFH_OPTIONS=xcd_swizzle=8 python scripts/asm_ab.py 60 2 asm_chunk_order 0 1
