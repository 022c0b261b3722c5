"""MFH_PRECOND_AUTO against both fixed choices over the rows of scripts/mg_robustness.py: iterations, solve time, the stretch the
library computed and what it chose; and the nearly incompressible rows with more Chebyshev steps on the fine levels.
    python scripts/r06/auto_precond_probe.py [grid]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
NAMES = {M.PRECOND_TWO_LEVEL: "two_level", M.PRECOND_MULTIGRID: "multigrid"}


def run(name, box, material, cells=None, extra=()):
    cx, cy, cz = cells or (n, n, n)
    V, T = grid.grid_tet_mesh(cx, cy, cz, [0, 0, 0], box)
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    material(c, V, T)
    big = 1e9
    c.bc_dirichlet_box([-1e-9, -big, -big], [1e-9, big, big], [0, 0, 0])
    c.bc_neumann_box([box[0] - 1e-9, -big, -big], [box[0] + 1e-9, big, big], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    rec = dict(case=name, grid=[cx, cy, cz], box=box)
    for pname, pc in (("two_level", M.PRECOND_TWO_LEVEL), ("multigrid", M.PRECOND_MULTIGRID), ("auto", M.PRECOND_AUTO)):
        c.set_preconditioner(pc)
        try:
            c.sim_solve(rtol=1e-8, maxit=20000)
            c.sim_solve(rtol=1e-8, maxit=20000)          # second solve: setup paid
            i = c.last_info
            rec[pname] = dict(iterations=i["iterations"], solve_ms=round(i["solve_ms"], 2))
            if pc == M.PRECOND_AUTO:
                k, a, st = c.precond_choice()
                rec[pname].update(chose=NAMES.get(k, k), stretch=round(st, 3))
        except M.MeshFEMHipError as e:
            rec[pname] = dict(error=str(e)[:80])
    for label, opts in extra:
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        for k, v in opts:
            c.set_option(k, v)
        try:
            c.sim_solve(rtol=1e-8, maxit=20000); c.sim_solve(rtol=1e-8, maxit=20000)
            rec[label] = dict(iterations=c.last_info["iterations"], solve_ms=round(c.last_info["solve_ms"], 2))
        except M.MeshFEMHipError as e:
            rec[label] = dict(error=str(e)[:80])
    best = min(rec[p]["solve_ms"] for p in ("two_level", "multigrid") if "solve_ms" in rec[p])
    if "solve_ms" in rec["auto"]:
        rec["auto_over_best"] = round(rec["auto"]["solve_ms"] / best, 3)
    print(json.dumps(rec), flush=True)
    c.close()


iso = lambda E, nu: (lambda c, V, T: c.material_isotropic(E, nu))


def contrast(ratio):
    def f(c, V, T):
        ctr = V[T].mean(axis=1)
        inside = ((ctr - ctr.mean(axis=0)) ** 2).sum(axis=1) < 0.08
        c.material_iso_field(np.where(inside, 200.0 * ratio, 200.0), np.full(len(T), 0.3))
    return f


steps = [("mg_steps_2", (("mg_steps_fine", 2), ("mg_steps_coarse", 2))), ("mg_steps_4", (("mg_steps_fine", 4), ("mg_steps_coarse", 4)))]
run("nu = 0.35 (headline material)", [1, 1, 1], iso(200.0, 0.35))
run("nu = 0.45", [1, 1, 1], iso(200.0, 0.45))
run("nu = 0.49", [1, 1, 1], iso(200.0, 0.49), extra=steps)
run("nu = 0.499", [1, 1, 1], iso(200.0, 0.499), extra=steps)
for s in (2, 4, 6, 8, 10, 12, 16):
    run("elements stretched %d : 1 : 1" % s, [s, 1, 1], iso(200.0, 0.35))
run("stretched 1 : 12 : 1 (another axis)", [1, 12, 1], iso(200.0, 0.35))
run("stiff inclusion, contrast 1e2", [1, 1, 1], contrast(1e2))
run("stiff inclusion, contrast 1e4", [1, 1, 1], contrast(1e4))
run("soft inclusion, contrast 1e-4", [1, 1, 1], contrast(1e-4))
run("reference bar_tet_2 (20 x 4 x 4 cells)", [5, 1, 1], iso(200.0, 0.35), cells=(20, 4, 4))
run("reference bar_tet_4 (80 x 16 x 16 cells)", [5, 1, 1], iso(200.0, 0.35), cells=(80, 16, 16))
