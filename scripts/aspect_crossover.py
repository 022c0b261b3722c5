"""Where does the two-level preconditioner overtake the multigrid V-cycle on stretched elements?   python scripts/aspect_crossover.py [grids...]
The unit grid scaled along x by the aspect; cantilever load; rtol 1e-8; iterations and solve time (second solve: hierarchy built) per preconditioner."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meshfem_amd as M
from meshfem_amd import grid

for n in [int(a) for a in sys.argv[1:]] or (24, 40):
    for asp in (1, 2, 3, 4, 6, 8, 16):
        V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [float(asp), 1, 1])
        rec = dict(grid=n, aspect=asp)
        for name, pre in (("multigrid", M.PRECOND_MULTIGRID), ("two_level", M.PRECOND_TWO_LEVEL)):
            c = M.Context(0)
            c.mesh_build(T, V, 2)
            c.material_isotropic(200.0, 0.35)
            c.bc_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9], [0, 0, 0])
            c.bc_neumann_box([asp - 1e-9, -9, -9], [asp + 1e-9, 9, 9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
            c.set_preconditioner(pre)
            try:
                c.sim_solve(rtol=1e-8, maxit=20000)
                c.sim_solve(rtol=1e-8, maxit=20000)
                rec[name] = dict(iterations=c.last_info["iterations"], solve_ms=round(c.last_info["solve_ms"], 2))
            except M.MeshFEMHipError as e:
                rec[name] = str(e)[:80]
            c.close()
        print(json.dumps(rec), flush=True)
