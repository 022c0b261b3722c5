"""Multigrid PCG away from the headline configuration: nearly incompressible materials, stretched elements, high-contrast per-element stiffness.
Iterations and times against the two-level preconditioner.       python scripts/mg_robustness.py [grid]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24


def run(name, box, material, cells=None, options=()):
    cx, cy, cz = cells or (n, n, n)
    V, T = grid.grid_tet_mesh(cx, cy, cz, [0, 0, 0], box)
    c = M.Context(0)
    for k, v in options:
        c.set_option(k, v)
    c.mesh_build(T, V, 2)
    material(c, V, T)
    big = 1e9
    c.bc_dirichlet_box([-1e-9, -big, -big], [1e-9, big, big], [0, 0, 0])
    c.bc_neumann_box([box[0] - 1e-9, -big, -big], [box[0] + 1e-9, big, big], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    out = {}
    for pname, pc in (("two_level", M.PRECOND_TWO_LEVEL), ("multigrid", M.PRECOND_MULTIGRID)):
        c.set_preconditioner(pc)
        try:
            u = c.sim_solve(rtol=1e-8, maxit=20000)
            i = c.last_info
            out[pname] = (i["iterations"], i["solve_ms"], i["true_rel_residual"], u)
        except M.MeshFEMHipError as e:
            out[pname] = ("FAILED: %s" % e, 0, 0, None)
    a, b = out["two_level"], out["multigrid"]
    diff = np.linalg.norm(a[3] - b[3]) / np.linalg.norm(a[3]) if a[3] is not None and b[3] is not None else float("nan")
    print("%-44s two-level %6s it %8.1f ms | multigrid %6s it %8.1f ms (res %.1e) | rel-L2 difference %.1e" % (name, a[0], a[1], b[0], b[1], b[2], diff), flush=True)
    c.close()


iso = lambda E, nu: (lambda c, V, T: c.material_isotropic(E, nu))


def contrast(ratio):
    def f(c, V, T):
        ctr = V[T].mean(axis=1)
        inside = ((ctr - ctr.mean(axis=0)) ** 2).sum(axis=1) < 0.08
        c.material_iso_field(np.where(inside, 200.0 * ratio, 200.0), np.full(len(T), 0.3))
    return f


run("nu = 0.35 (headline material)", [1, 1, 1], iso(200.0, 0.35))
run("nu = 0.45", [1, 1, 1], iso(200.0, 0.45))
run("nu = 0.49", [1, 1, 1], iso(200.0, 0.49))
run("nu = 0.499", [1, 1, 1], iso(200.0, 0.499))
run("elements stretched 4 : 1 : 1", [4, 1, 1], iso(200.0, 0.35))
run("elements stretched 16 : 1 : 1", [16, 1, 1], iso(200.0, 0.35))
run("stiff inclusion, contrast 1e2", [1, 1, 1], contrast(1e2))
run("stiff inclusion, contrast 1e4", [1, 1, 1], contrast(1e4))
run("soft inclusion, contrast 1e-4", [1, 1, 1], contrast(1e-4))
# the reference's own cantilever bars (examples/cantilever/gen.sh:5: grid 5 2^i x 2^i x 2^i, cubic cells): a slender DOMAIN of isotropic elements
run("reference bar_tet_2 (20 x 4 x 4 cells)", [5, 1, 1], iso(200.0, 0.35), cells=(20, 4, 4))
run("reference bar_tet_4 (80 x 16 x 16 cells)", [5, 1, 1], iso(200.0, 0.35), cells=(80, 16, 16))
# stretched elements again with the bins of the aggregate levels in the elements' proportions / a stronger fine smoother
run("16 : 1 : 1, mg_anisotropic_bins 1", [16, 1, 1], iso(200.0, 0.35), options=(("mg_anisotropic_bins", 1),))
run("16 : 1 : 1, 4 fine steps on [0.02, 1] lambda_max", [16, 1, 1], iso(200.0, 0.35), options=(("mg_steps_fine", 4), ("mg_ratio_fine", 0.02), ("mg_steps_coarse", 4), ("mg_ratio_coarse", 0.02)))
run("16 : 1 : 1, 8 fine steps on [0.005, 1] lambda_max", [16, 1, 1], iso(200.0, 0.35), options=(("mg_steps_fine", 8), ("mg_ratio_fine", 0.005), ("mg_steps_coarse", 8), ("mg_ratio_coarse", 0.005)))
