"""The device arena under the sequence the bench line runs in ONE process: contexts of other sizes first (60^3 quadratic with a multigrid
solve, the same with both triangles stored, 35^3 linear), all closed, then the one-shot first assembly + first multigrid solve of the 119^3
cube -- the leg that took 4.8 s on the round-4 driver box (size-bucket cache: 4.5 s of hipMalloc in the symbolic phase) against 0.6-0.85 s
in a fresh process.     python scripts/arena_probe.py [big grid] [fresh]      ("fresh": skip the warm-up contexts)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid

big = int(sys.argv[1]) if len(sys.argv) > 1 else 119
fresh = len(sys.argv) > 2 and sys.argv[2] == "fresh"
out = dict(big_grid=big, fresh_process=fresh, stages=[])


def gb(st):
    return {k: (round(v / 1e9, 2) if k.endswith("bytes") else v) for k, v in st.items()}


def run(n, deg, storage=None, solve=True, label=""):
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    T = np.ascontiguousarray(T, dtype=np.int32)
    c = M.Context(0)
    if storage is not None:
        c.set_option("matrix_storage", storage)
    t0 = time.perf_counter(); c.mesh_build(T, V, deg); t1 = time.perf_counter()
    del V, T
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    c.symbolic(False); t2 = time.perf_counter()
    c.assemble(); c.dev_sync(); t3 = time.perf_counter()
    rec = dict(label=label, grid=n, deg=deg, elements=int(c.n_elem), mesh_build_s=t1 - t0, symbolic_s=t2 - t1, first_pass_s=t3 - t2,
               first_assembly_s=t3 - t0)
    if solve:
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        t0 = time.perf_counter()
        c.sim_solve(rtol=1e-8, maxit=2000)
        rec.update(first_solve_wall_s=time.perf_counter() - t0, iterations=c.last_info["iterations"], solve_s=c.last_info["solve_ms"] * 1e-3,
                   hierarchy_setup_s=c.multigrid_info()["setup_ms"] * 1e-3)
    rec["arena_live"] = gb(M.device_arena_stats(0))
    c.close()
    rec["arena_closed"] = gb(M.device_arena_stats(0))
    rec["cache"] = M.device_cache_stats(0)
    out["stages"].append(rec)
    print(json.dumps(rec), flush=True)


if not fresh:
    run(60, 2, label="configs[2]")
    run(60, 2, storage=0, label="configs[2], both triangles")
    run(35, 1, label="configs[1]")
    run(44, 2, label="44^3 quadratic")
run(big, 2, label="configs[4] cube, one-shot")
run(big, 2, label="configs[4] cube, again")
print(json.dumps(out))
