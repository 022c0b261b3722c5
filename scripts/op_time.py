"""Per-iteration time of a fixed-length classic PCG run and of the operator alone at a given grid (A/B builds on one box).
    python scripts/op_time.py [grid] [iters] [precond]"""
import sys

sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
pre = int(sys.argv[3]) if len(sys.argv) > 3 else 0
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
for geov in (1, 0):
    c.set_option("mf_geometry_from_vertices", geov)
    ops = [c.time_spmv_kernel(50) for _ in range(3)]
    its = []
    for rep in range(3):
        try:
            c.solve(f, rtol=1e-30, maxit=iters)
        except M.MeshFEMHipError:
            pass
        its.append(c.last_info["solve_ms"] / iters)
    print("geometry from vertices %d: operator alone %.4f ms (min of 3), PCG iteration %.4f ms (min of 3)" % (geov, min(ops), min(its)), flush=True)
