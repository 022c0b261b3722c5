"""Linear tets: assembled block-CSR SpMV (k_spmv, both triangles) against the matrix-free cluster operator (option matrix_free 1), operator alone and the
block-Jacobi PCG iteration.    python scripts/p1_mf_probe.py [grid ...]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

for n in [int(a) for a in sys.argv[1:]] or [35, 64]:
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    T = np.ascontiguousarray(T, dtype=np.int32)
    for mf in (0, 1):
        c = M.Context(0)
        c.set_option("matrix_free", mf)
        c.mesh_build(T, V, 1)
        c.material_isotropic(200.0, 0.35)
        c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
        c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
        c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
        try:
            c.sim_solve(rtol=1e-30, maxit=300)
        except M.MeshFEMHipError:
            pass
        it = c.last_info["solve_ms"] / 300
        op = min(c.time_spmv_kernel(50) for _ in range(3))
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u = c.sim_solve(rtol=1e-8, maxit=500)
        print("%d^3 P1 (%d tets), matrix_free %d: operator %.4f ms, PCG block-Jacobi iteration %.4f ms, multigrid solve %.2f ms (%d iterations), max|u| %.10g; %s"
              % (n, len(T), mf, op, it, c.last_info["solve_ms"], c.last_info["iterations"], np.abs(u).max(), c.matrix_free_info() if mf else ""), flush=True)
        c.close()
