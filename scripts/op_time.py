"""Per-iteration time of fixed-length PCG runs at a given grid, with and without the operator's geometry recomputation.
    python scripts/op_time.py [grid] [nrhs] [iters] [precond]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
nrhs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
pre = int(sys.argv[4]) if len(sys.argv) > 4 else 0
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.assemble()
f = c.neumann_load().ravel()
vars_, vals = c.bc_dirichlet_vars()
c.fix_variables(vars_, vals)
c.set_preconditioner(pre)
c.set_option("batch_rhs", 1 if nrhs > 1 else 0)
F = np.stack([f * (1 + k) for k in range(nrhs)])
for geov in (1, 0):
    c.set_option("mf_geometry_from_vertices", geov)
    print("operator alone (geometry from vertices %d): %.4f ms" % (geov, c.time_spmv_kernel(50)), flush=True)
    for variant in (1, 0):
        c.set_option("pcg_variant", variant)
        for rep in range(2):
            try:
                c.solve_batch(F, rtol=1e-30, maxit=iters)
            except M.MeshFEMHipError:
                pass
        i = c.last_infos
        tot = sum(x["solve_ms"] for x in i) if variant == 0 or nrhs == 1 else i[0]["solve_ms"]
        print("  geov %d variant %d nrhs %d: %.3f ms per iteration (all rhs), %.3f per rhs" % (geov, variant, nrhs, tot / iters, tot / iters / nrhs), flush=True)
