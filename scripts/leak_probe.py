"""Device memory after repeated context life cycles (mesh build, assembly, hierarchy, solves with every preconditioner, destroy): the free
memory the driver reports must come back.     python scripts/leak_probe.py [cycles]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import meshfem_amd as M
from meshfem_amd import grid
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 12
V, T = grid.grid_tet_mesh(20, 20, 20, [0, 0, 0], [1, 1, 1])
free0 = None
for k in range(cycles):
    c = M.Context(0)
    c.mesh_build(T, V, 2 if k % 3 else 1)
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    for pre in (M.PRECOND_MULTIGRID, M.PRECOND_TWO_LEVEL, M.PRECOND_BLOCK_JACOBI):
        c.set_preconditioner(pre)
        c.sim_solve(rtol=1e-6)
    c.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(0)
    if k == 1:
        free0 = free          # after the first cycles the runtime's own pools have settled
    print("cycle %2d: free %.3f GB" % (k, free / 1e9), flush=True)
print("drift after cycle 1: %.1f MB" % ((free0 - free) / 1e6))

# one context, the hierarchy rebuilt again and again (new fixed variables invalidate it), vertex updates in between
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.set_preconditioner(M.PRECOND_MULTIGRID)
rng = np.random.default_rng(0)
f1 = None
for k in range(cycles):
    c.bc_clear()
    c.bc_dirichlet_box([-1e-3, -1e9, -1e9], [1e-3 + 0.05 * (k % 3), 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-3, -1e9, -1e9], [1 + 1e-3, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    c.sim_solve(rtol=1e-6)
    c.mesh_update_vertices(V + 1e-5 * rng.standard_normal(V.shape))
    c.sim_solve(rtol=1e-6)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(0)
    if k == 1:
        f1 = free
    print("rebuild %2d: free %.3f GB, iterations %d" % (k, free / 1e9, c.last_info["iterations"]), flush=True)
print("drift after rebuild 1: %.1f MB" % ((f1 - free) / 1e6))
c.close()
