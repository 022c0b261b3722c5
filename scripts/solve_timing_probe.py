"""Host-side laps of Simulator::solve at the bench size: run with MFH_SOLVE_TIMING=1 (two solves: first with setup)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, meshfem_amd as M
from meshfem_amd import grid
V, T = grid.grid_tet_mesh(60, 60, 60, [0, 0, 0], [1, 1, 1])
c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(200., 0.35)
c.bc_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9], [0, 0, 0]); c.bc_neumann_box([1 - 1e-9, -9, -9], [1 + 1e-9, 9, 9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.set_preconditioner(M.PRECOND_TWO_LEVEL); c.assemble()
for k in range(2):
    t0 = time.time(); u = c.sim_solve(); print("solve wall %.3f s, info %s" % (time.time() - t0, {k2: c.last_info[k2] for k2 in ("iterations", "solve_ms", "setup_ms")}), flush=True)
