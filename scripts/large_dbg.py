import sys, time
sys.path.insert(0, ".")
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid
n = int(sys.argv[1])
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.assemble(); print("assembled", flush=True)
for geov in (0, 1):
    c.set_option("mf_geometry_from_vertices", geov)
    t0 = time.time(); ms = c.time_spmv_kernel(3); print("mf operator geov", geov, ms, "ms, wall", time.time() - t0, flush=True)
print(c.matrix_free_info(), flush=True)
f = c.neumann_load().ravel(); print("load", flush=True)
vars_, vals = c.bc_dirichlet_vars(); c.fix_variables(vars_, vals); print("fixed", len(vars_), flush=True)
t0 = time.time()
try:
    c.solve(f, rtol=1e-30, maxit=100)
except M.MeshFEMHipError as e:
    print("expected:", str(e)[:80])
print("BJ 100 iterations", c.last_info["solve_ms"], "ms, wall", time.time() - t0, flush=True)
c.set_preconditioner(M.PRECOND_TWO_LEVEL)
t0 = time.time()
try:
    c.solve(f, rtol=1e-30, maxit=100)
except M.MeshFEMHipError as e:
    print("expected:", str(e)[:80])
print("TL 100 iterations", c.last_info["solve_ms"], "ms, wall", time.time() - t0, c.precond_info(), flush=True)
