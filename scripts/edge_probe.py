"""Edge cases of the solve path that must neither crash nor hang: one element, all variables fixed, zero load, two components, a one-layer slab;\nevery preconditioner.     python scripts/edge_probe.py"""
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import meshfem_amd as M
from meshfem_amd import grid

def run(name, fn):
    try:
        print(name, "->", fn(), flush=True)
    except M.MeshFEMHipError as e:
        print(name, "-> MeshFEMHipError:", str(e)[:160], flush=True)

def one_tet(deg, pre):
    V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]]); T = np.array([[0, 1, 2, 3]])
    c = M.Context(0); c.mesh_build(T, V, deg); c.material_isotropic(1.0, 0.3)
    c.fix_variables(np.arange(9)); c.set_preconditioner(pre)
    f = np.zeros(3 * c.n_node); f[9:12] = [0, 0, -1.0]
    u = c.solve(f, rtol=1e-10); return c.last_info["iterations"], float(np.abs(u).max())

def two_tris(deg, pre):
    V = np.array([[0, 0], [1, 0], [1, 1], [0, 1.0]]); T = np.array([[0, 1, 2], [0, 2, 3]])
    c = M.Context(0); c.mesh_build(T, V, deg); c.material_isotropic(1.0, 0.3)
    c.fix_variables(np.array([0, 1, 2, 3])); c.set_preconditioner(pre)
    f = np.zeros(2 * c.n_node); f[4:6] = [0, -1.0]
    u = c.solve(f, rtol=1e-10); return c.last_info["iterations"], float(np.abs(u).max())

def all_fixed(pre):
    V, T = grid.grid_tet_mesh(3, 3, 3, [0, 0, 0], [1, 1, 1])
    c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(1.0, 0.3)
    c.fix_variables(np.arange(3 * c.n_node)); c.set_preconditioner(pre)
    u = c.solve(np.ones(3 * c.n_node), rtol=1e-10); return c.last_info["iterations"], float(np.abs(u).max())

def zero_rhs(pre):
    V, T = grid.grid_tet_mesh(4, 4, 4, [0, 0, 0], [1, 1, 1])
    c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(1.0, 0.3)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0]); c.set_preconditioner(pre)
    u = c.sim_solve(rtol=1e-8); return c.last_info["iterations"], float(np.abs(u).max())

def two_components(pre):
    V1, T1 = grid.grid_tet_mesh(4, 4, 4, [0, 0, 0], [1, 1, 1]); V2 = V1 + [3.0, 0, 0]
    V = np.vstack([V1, V2]); T = np.vstack([T1, T1 + len(V1)])
    c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(1.0, 0.3)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0]); c.bc_dirichlet_box([3 - 1e-9, -1e9, -1e9], [3 + 1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    c.set_preconditioner(pre)
    u = c.sim_solve(rtol=1e-8); return c.last_info["iterations"], c.last_info["true_rel_residual"], c.precond_info().get("note", "")[:80]

def thin_slab(pre):     # one element layer: lattice aggregates degenerate in z
    V, T = grid.grid_tet_mesh(40, 40, 1, [0, 0, 0], [1, 1, 0.025])
    c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(1.0, 0.3)
    c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
    c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, 0, -1], kind=M.NEUMANN_TRACTION)
    c.set_preconditioner(pre)
    u = c.sim_solve(rtol=1e-8, maxit=20000); return c.last_info["iterations"], c.last_info["true_rel_residual"], c.precond_info().get("note", "")[:80]

for pre, pn in ((M.PRECOND_MULTIGRID, "mg"), (M.PRECOND_TWO_LEVEL, "tl"), (M.PRECOND_BLOCK_JACOBI, "bj")):
    for deg in (1, 2):
        run("one_tet deg%d %s" % (deg, pn), lambda: one_tet(deg, pre))
        run("two_tris deg%d %s" % (deg, pn), lambda: two_tris(deg, pre))
    run("all_fixed %s" % pn, lambda: all_fixed(pre))
    run("zero_rhs %s" % pn, lambda: zero_rhs(pre))
    run("two_components %s" % pn, lambda: two_components(pre))
    run("thin_slab %s" % pn, lambda: thin_slab(pre))
print("DONE")


def all_fixed_nonzero(pre):
    V, T = grid.grid_tet_mesh(3, 3, 3, [0, 0, 0], [1, 1, 1])
    c = M.Context(0); c.mesh_build(T, V, 2); c.material_isotropic(1.0, 0.3)
    vals = np.linspace(-1, 1, 3 * c.n_node)
    c.fix_variables(np.arange(3 * c.n_node), vals); c.set_preconditioner(pre)
    u = c.solve(np.ones(3 * c.n_node), rtol=1e-10); return c.last_info["iterations"], float(np.abs(u - vals).max())

def rebuild_same_context(pre):
    c = M.Context(0); c.set_preconditioner(pre); out = []
    for n, deg in ((6, 2), (3, 1), (8, 2), (5, 1)):
        V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
        c.mesh_build(T, V, deg); c.material_isotropic(1.0, 0.3)
        c.bc_dirichlet_box([-1e-9, -1e9, -1e9], [1e-9, 1e9, 1e9], [0, 0, 0])
        c.bc_neumann_box([1 - 1e-9, -1e9, -1e9], [1 + 1e-9, 1e9, 1e9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
        c.sim_solve(rtol=1e-8); out.append((c.last_info["iterations"], "%.1e" % c.last_info["true_rel_residual"]))
    return out

for pre, pn in ((M.PRECOND_MULTIGRID, "mg"), (M.PRECOND_TWO_LEVEL, "tl"), (M.PRECOND_BLOCK_JACOBI, "bj")):
    run("all_fixed_nonzero %s" % pn, lambda: all_fixed_nonzero(pre))
    run("rebuild_same_context %s" % pn, lambda: rebuild_same_context(pre))
print("DONE 2")
