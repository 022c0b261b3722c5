"""Option mg_fuse on / off: configs[2]-shaped solve (60^3, isotropic) (the configs[3] cell problems: MFH_OPTIONS="mg_fuse=0" python bench.py --leg config3, scripts/r06/run_z.sh).
python scripts/r06/fuse_probe.py"""
import os, sys, time
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
c.set_preconditioner(M.PRECOND_MULTIGRID)
u = c.sim_solve(rtol=1e-8, maxit=800)
for rep in range(3):
    for fuse, f32 in ((1, 1), (1, 0), (0, 0)):
        c.set_option("mg_fuse", fuse)
        c.set_option("mg_dinv_fp32", f32)
        u = c.sim_solve(rtol=1e-8, maxit=800)
        i = c.last_info
        print("configs[2] shape  mg_dinv_fp32 %d" % f32, end="  ")
        print("mg_fuse %d: %d iterations, solve %.2f ms (%.3f ms / iteration), true residual %.2e, max|u| %.12g"
              % (fuse, i["iterations"], i["solve_ms"], i["solve_ms"] / max(1, i["iterations"]), i["true_rel_residual"], np.abs(u).max()), flush=True)
c.close()

