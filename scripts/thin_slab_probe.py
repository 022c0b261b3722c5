"""Block-Jacobi PCG on a one-layer plate in bending (40 x 40 x 1 cells, 1 : 40 aspect): how many iterations the plain preconditioner needs when
the stagnation check is out of the way (check_every 2500 -> window 100 000).     python scripts/thin_slab_probe.py"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meshfem_amd as M
from meshfem_amd import grid
V, T = grid.grid_tet_mesh(40, 40, 1, [0, 0, 0], [1, 1, 0.025])
c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(1.0, 0.3)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, 0, -1], kind=M.NEUMANN_TRACTION)
c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
c.set_option("check_every", 2500)
t0 = time.time()
try:
    u = c.sim_solve(rtol=1e-8, maxit=100000)
    print("converged:", c.last_info, time.time() - t0)
except M.MeshFEMHipError as e:
    print("error:", e, c.last_info, time.time() - t0)
