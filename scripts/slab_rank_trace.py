"""What ONE rank's GPU executes per multigrid iteration when configs[4]'s cube is dealt out over 8 ranks: the rank's slab (119 x 119 x 15 hex layers: an
eighth of the cube) solved as ONE unpartitioned context -- the same element count, the same levels (an eighth of the aggregates each), no halos -- for
`rocprofv3 --kernel-trace`.        python scripts/slab_rank_trace.py [n] [layers]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 119
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 15
V, T = grid.grid_tet_mesh(n, n, layers, [0, 0, 0], [1, 1, layers / float(n)])
T = np.ascontiguousarray(T, dtype=np.int32)
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.set_preconditioner(M.PRECOND_MULTIGRID)
u = c.sim_solve(rtol=1e-8, maxit=500)
i = c.last_info
print("slab %d x %d x %d: %d P2 tets, %d DOF; multigrid %d iterations, %.1f ms = %.3f ms per iteration; hierarchy %.0f ms; levels %s"
      % (n, n, layers, c.n_elem, 3 * c.n_dof, i["iterations"], i["solve_ms"], i["solve_ms"] / i["iterations"], c.multigrid_info()["setup_ms"],
         [L["aggregates"] for L in c.multigrid_levels()]), flush=True)
