"""Same-process A/B of runtime options: operator alone and PCG iteration (min of 3), classic loop (pcg_variant 0) unless PCGV is set.
    python scripts/op_opts.py grid option v1 v2 ...      e.g.  python scripts/op_opts.py 60 mf_lane_stride 1 37 1
    PCGV=1 python scripts/op_opts.py 60 pcg_variant 1 0   (Chronopoulos-Gear vs classic)"""
import os
import sys

sys.path.insert(0, ".")
import meshfem_amd as M
from meshfem_amd import grid

n, opt, vals = int(sys.argv[1]), sys.argv[2], [float(v) for v in sys.argv[3:]]
pre = int(os.environ.get("PRECOND", "0"))
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
c = M.Context(0)
c.mesh_build(T, V, 2)
c.material_isotropic(200.0, 0.35)
c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
c.assemble()
f = c.neumann_load().ravel()
vars_, vals_ = c.bc_dirichlet_vars()
c.fix_variables(vars_, vals_)
c.set_preconditioner(pre)
if "PCGV" in os.environ:
    c.set_option("pcg_variant", int(os.environ["PCGV"]))
for v in vals:
    c.set_option(opt, v)
    ops = [c.time_spmv_kernel(50) for _ in range(3)]
    its = []
    for rep in range(3):
        try:
            c.solve(f, rtol=1e-30, maxit=300)
        except M.MeshFEMHipError:
            pass
        its.append(c.last_info["solve_ms"] / 300)
    print("%s = %g: operator alone %.4f ms, PCG iteration %.4f ms" % (opt, v, min(ops), min(its)), flush=True)
