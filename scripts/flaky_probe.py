import sys
sys.path.insert(0, ".")
import numpy as np
import meshfem_amd as M
from oracle import meshfem_oracle as O
V, T = O.grid_tet_mesh(3, 2, 2)
out = {}
for rep in range(60):
    c = M.Context(0)
    c.mesh_build(T, V, 1)
    c.material_isotropic(200.0, 0.35)
    c.bc_neumann_box([3 - 1e-9, -9, -9], [3 + 1e-9, 9, 9], [1.0, 0.5, 0.0], kind=M.NEUMANN_TRACTION)
    try:
        u = c.sim_solve_constrained(flags=M.SOLVE_ALLOW_ILL_POSED, maxit=100000)
        key = "RETURNED: %s max|u| %.3g" % ({k: c.last_info[k] for k in ("iterations", "converged", "rel_residual", "true_rel_residual", "used_graph")}, np.abs(u).max())
    except M.MeshFEMHipError as e:
        key = "raised: " + str(e)[:60]
    out[key] = out.get(key, 0) + 1
    c.close()
for k, v in out.items():
    print(v, k)
