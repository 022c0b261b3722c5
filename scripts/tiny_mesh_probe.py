"""Meshes of one element block and a handful of blocks (no or few interface rows): the matrix-free cluster operator against the assembled matrix,
block-Jacobi and multigrid solves.    python scripts/tiny_mesh_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
for dims in [(1, 1, 1), (2, 2, 2), (3, 2, 1), (5, 4, 3)]:
    V, T = grid.grid_tet_mesh(*dims)
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.material_isotropic(200.0, 0.35)
    c.assemble()
    A = c.export_scipy()
    x = np.random.default_rng(0).standard_normal(A.shape[0])
    y = c.apply_K(x)
    info = c.matrix_free_info() if hasattr(c, "matrix_free_info") else None
    err = np.abs(y - A @ x).max() / np.abs(A @ x).max()
    print(dims, len(T), "tets: matrix-free vs assembled rel err %.2e" % err, info)
    assert err < 1e-13
    c.bc_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9], [0, 0, 0])
    c.bc_neumann_box([dims[0] - 1e-9, -9, -9], [dims[0] + 1e-9, 9, 9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    for pre in (M.PRECOND_BLOCK_JACOBI, M.PRECOND_MULTIGRID):
        c.set_preconditioner(pre)
        u = c.sim_solve(rtol=1e-10)
        print("   precond", pre, c.last_info["iterations"], c.last_info["converged"], c.precond_info().get("note", ""))
        assert c.last_info["converged"]
    c.close()
print("ok")
