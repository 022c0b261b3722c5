"""Where the wall time of a periodic homogenization run goes (BASELINE configs[3] at full size, multigrid PCG): cProfile of
meshfem_amd.homogenization.homogenize, top entries by cumulative and by own time.     python scripts/hom_profile.py [grid]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import meshfem_amd as M
from meshfem_amd import grid, homogenization as H

n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
P = grid.synthetic_orthotropic_field(len(T), 3, 0)
for rep in range(2):
    pr = cProfile.Profile()
    t0 = time.time()
    pr.enable()
    r = H.homogenize(V, T, 2, ortho_params=P, rtol=1e-8, preconditioner=M.PRECOND_MULTIGRID)
    pr.disable()
    print("rep %d: wall %.3f s, iterations %s, solve_ms %s, timing %s" % (rep, time.time() - t0, r["iterations"], [round(i["solve_ms"], 1) for i in r["infos"]],
                                                                         r["sim"].ctx.timing()), flush=True)
    if rep == 1:
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(28)
        st.sort_stats("tottime").print_stats(14)
    r["sim"].ctx.close()
