"""EXPERIMENT RECORD (the MFH_GRID_CAP switch it drove was removed again: the cap does not matter, profiles/r06_vector_kernels_two_states.txt).
Cap of the grid-stride vector kernels that end in a reduction (MFH_GRID_CAP workgroups of 256 lanes): block-Jacobi PCG and multigrid PCG at configs[2]'s shape.
python scripts/r06/grid_cap_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = 60
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.assemble()
cap = os.environ.get("MFH_GRID_CAP", "2048")
for pre, name, maxit in ((M.PRECOND_MULTIGRID, "multigrid", 800), (M.PRECOND_BLOCK_JACOBI, "block-Jacobi", 300)):
    c.set_preconditioner(pre)
    for rep in range(3):
        try:
            u = c.sim_solve(rtol=1e-8, maxit=maxit)
        except RuntimeError:
            pass
        i = c.last_info
    print("cap %6s  %-13s %4d iterations  %.4f ms / iteration" % (cap, name, i["iterations"], i["solve_ms"] / max(1, i["iterations"])), flush=True)
