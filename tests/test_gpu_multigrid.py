"""MFH_PRECOND_MULTIGRID: the p-multigrid V-cycle (quadratic level -> linear level on the same vertices -> rigid-body modes of
aggregates; mfh_multigrid.cpp) as the preconditioner of the PCG that replaces the reference's CHOLMOD solve
(SPSDSystem::solve, SparseMatrices.hh:2515-2606). The answers are those of the oracle's direct solve (north_star tolerance 1e-6
rel-L2 on nodal displacements; measured ~1e-9 at rtol 1e-10); the iteration counts must be mesh-independent tens."""
import numpy as np
import pytest

import meshfem_amd as M
from meshfem_amd import grid
from oracle import meshfem_oracle as O

pytestmark = pytest.mark.gpu

U_RTOL = 1e-6


def _cantilever(n, deg=2, dim=3):
    if dim == 3:
        V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    else:
        V, Q = O.gen_grid_2d(n, n)
        V, T = O.quad_tri_subdiv(V, Q)
        V = V[:, :2] / n
    c = M.Context(0)
    c.mesh_build(T, V, deg)
    c.material_isotropic(200.0, 0.35)
    lo, hi = [-9.0] * dim, [9.0] * dim
    lo[0], hi[0] = -1e-9, 1e-9
    c.bc_dirichlet_box(lo, hi, [0.0] * dim)
    lo2, hi2 = [-9.0] * dim, [9.0] * dim
    lo2[0], hi2[0] = 1 - 1e-9, 1 + 1e-9
    c.bc_neumann_box(lo2, hi2, [0.0, -1.0, 0.0][:dim], kind=M.NEUMANN_TRACTION)
    return c, V, T


@pytest.mark.parametrize("dim,n", [(3, 4), (2, 12)])
def test_multigrid_pcg_matches_the_oracle_direct_solve(dim, n):
    c, V, T = _cantilever(n, 2, dim)
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    u = c.sim_solve(rtol=1e-10)
    i = c.last_info
    g = c.multigrid_info()
    assert i["converged"] and i["true_rel_residual"] < 2e-10
    assert g["fine_dof"] == c.n_dof and 0 < g["coarse_dof"] < g["fine_dof"] and g["lambda_max_fine"] > 1.0
    sim = O.Simulator(T, V, 2)
    sim.set_material_constant(O.ElasticityTensor.isotropic(dim, 200.0, 0.35))
    lo, hi = [-9.0] * dim, [9.0] * dim
    lo[0], hi[0] = -1e-9, 1e-9
    sim.apply_dirichlet_box(lo, hi, [0.0] * dim)
    lo[0], hi[0] = 1 - 1e-9, 1 + 1e-9
    sim.apply_neumann_box(lo, hi, [0.0, -1.0, 0.0][:dim], "traction")
    uref = sim.solve()
    assert np.linalg.norm(u - uref) / np.linalg.norm(uref) < U_RTOL
    # the same system through the two-level preconditioner: same answer, more iterations
    c.set_preconditioner(M.PRECOND_TWO_LEVEL)
    u2 = c.sim_solve(rtol=1e-10)
    assert np.linalg.norm(u2 - u) <= 1e-7 * np.linalg.norm(u)
    assert i["iterations"] < c.last_info["iterations"]
    c.close()


def test_multigrid_iteration_count_does_not_grow_with_the_mesh():
    its = {}
    for n in (6, 12):
        c, _, _ = _cantilever(n)
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        c.sim_solve(rtol=1e-8)
        its[n] = c.last_info["iterations"]
        c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
        c.sim_solve(rtol=1e-8)
        bj = c.last_info["iterations"]
        c.close()
        assert its[n] < 0.2 * bj, (n, its[n], bj)
    assert its[12] <= 1.3 * its[6] + 3 and its[12] <= 60, its      # eight times the elements, (about) the same count


def test_multigrid_nonzero_dirichlet_values_options_and_graph_free_loop():
    c, V, T = _cantilever(5)
    pos = c.node_positions()
    right = np.flatnonzero(np.abs(pos[:, 0] - 1) < 1e-12)
    c.assemble()
    vars_, vals = c.bc_dirichlet_vars()
    extra = (3 * right[:, None] + np.arange(3)[None, :]).ravel()
    c.fix_variables(np.concatenate([vars_, extra]), np.concatenate([vals, np.tile([0.01, 0.0, -0.02], len(right))]))
    f = np.zeros(3 * c.n_dof)
    c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
    u_bj = c.solve(f, rtol=1e-11)
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    u = c.solve(f, rtol=1e-11)
    it = c.last_info["iterations"]
    assert np.linalg.norm(u - u_bj) <= 1e-8 * np.linalg.norm(u_bj)
    assert np.allclose(u.reshape(-1, 3)[right], [0.01, 0.0, -0.02], atol=1e-15)
    for opt, val in (("mg_steps_fine", 2), ("mg_steps_coarse", 2), ("mg_coarse_cycles", 2), ("pcg_graph", 0)):
        c.set_option(opt, val)
        u2 = c.solve(f, rtol=1e-11)
        assert c.last_info["converged"] and np.linalg.norm(u2 - u) <= 1e-8 * np.linalg.norm(u), opt
    assert c.last_info["used_graph"] == 0 and it < 60
    c.close()


@pytest.mark.parametrize("deg,n", [(2, 6), (1, 16)])
def test_multigrid_orthotropic_field_and_periodic_cell_problems(deg, n):
    """BASELINE configs[3] in small: per-element orthotropic field, periodic DoF map, pinned node, six cell problems. Linear elements
    (round 5: matrix-free operator, upper-triangle storage, the Galerkin product of the periodic lattice completed from the stored
    triangle) on a grid whose lattice has more than two bins per axis."""
    from meshfem_amd import homogenization as H
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    P = grid.synthetic_orthotropic_field(len(T), 3, seed=1)
    res = {}
    for pre in (M.PRECOND_TWO_LEVEL, M.PRECOND_MULTIGRID):
        sim = M.Simulator(T, V, deg)
        sim.ctx.material_ortho_field(P)
        sim.ctx.set_preconditioner(pre)
        sim.rtol = 1e-10
        w, infos = H.solve_cell_problems(sim)
        Ch = H.homogenized_elasticity_tensor(sim, w)
        assert all(i["converged"] for i in infos)
        res[pre] = (np.asarray(Ch), max(i["iterations"] for i in infos), sim.ctx.precond_info()["note"], sim.ctx.multigrid_levels() if pre == M.PRECOND_MULTIGRID else None)
        sim.ctx.close()
    assert np.abs(res[M.PRECOND_MULTIGRID][0] - res[M.PRECOND_TWO_LEVEL][0]).max() <= 1e-7 * np.abs(res[M.PRECOND_TWO_LEVEL][0]).max()
    assert res[M.PRECOND_MULTIGRID][1] < res[M.PRECOND_TWO_LEVEL][1], res


@pytest.mark.parametrize("dim,n", [(3, 8), (2, 40)])
def test_multigrid_aggregate_hierarchy_levels_and_the_dense_only_variant(dim, n):
    """Below the linear level: rigid-body modes of lattice bins in stencil storage, merged 2^dim at a time, dense inverse at the end.
    Small `mg_agg_target` / `mg_dense_max` force several stencil levels on a small mesh (3D and 2D code paths of the k_st_* kernels);
    `mg_agg_target 0` switches the hierarchy off (the linear context's own ~1000-aggregate dense coarse space). Same solution every way."""
    c, V, T = _cantilever(n, 2, dim)
    c.set_preconditioner(M.PRECOND_TWO_LEVEL)
    u_ref = c.sim_solve(rtol=1e-11)
    it_tl = c.last_info["iterations"]
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    res = {}
    for name, opts in (("default", {}), ("deep", {"mg_agg_target": 6, "mg_dense_max": 8}), ("dense_only", {"mg_agg_target": 0}),
                       ("no_over_correction", {"mg_over_correction": 1.0})):
        c.set_option("mg_agg_target", 32); c.set_option("mg_dense_max", 1200); c.set_option("mg_over_correction", 1.5)
        for k, v in opts.items():
            c.set_option(k, v)
        u = c.sim_solve(rtol=1e-11)
        i, p = c.last_info, c.precond_info()
        assert i["converged"] and np.linalg.norm(u - u_ref) <= 1e-8 * np.linalg.norm(u_ref), name
        res[name] = (i["iterations"], p["aggregates"], p["coarse_dim"])
    nm = 6 if dim == 3 else 3
    assert res["deep"][2] <= 8 * nm and res["deep"][1] > res["default"][1]          # more, smaller aggregates; tiny dense level
    assert max(r[0] for r in res.values()) < it_tl, (res, it_tl)
    c.close()


def test_multigrid_on_a_caller_supplied_node_table():
    """mfh_mesh_set (any node numbering, no topology): the transfer lists come from the host loops over the element table instead of
    the device route of the library's own numbering; same solution as block-Jacobi."""
    n = 5
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    h = M.Context(-1)
    h.mesh_build(T, V, 2)
    en, pos = h.elem_nodes(), h.node_positions()
    h.close()
    perm = np.random.default_rng(5).permutation(len(pos))          # new id of old node k: perm[k]
    pos2 = np.empty_like(pos); pos2[perm] = pos
    c = M.Context(0)
    c.mesh_set(3, 2, perm[en].astype(np.int32), pos2)
    c.material_isotropic(200.0, 0.35)
    fixed = np.flatnonzero(np.abs(pos2[:, 0]) < 1e-12)
    c.fix_variables((3 * fixed[:, None] + np.arange(3)[None, :]).ravel())
    f = np.zeros((len(pos2), 3)); f[np.abs(pos2[:, 0] - 1) < 1e-12, 1] = -1e-3
    c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
    u_bj = c.solve(f.ravel(), rtol=1e-11)
    it_bj = c.last_info["iterations"]
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    u = c.solve(f.ravel(), rtol=1e-11)
    assert c.last_info["converged"] and c.multigrid_info()["coarse_dof"] == len(V)
    assert np.linalg.norm(u - u_bj) <= 1e-8 * np.linalg.norm(u_bj) and c.last_info["iterations"] < 0.3 * it_bj
    c.close()


@pytest.mark.parametrize("dim,n", [(3, 6), (2, 24)])
def test_multigrid_on_linear_elements_matches_the_oracle(dim, n):
    """Linear elements enter the hierarchy at its linear level: Chebyshev on the context's own assembled K, aggregates below."""
    c, V, T = _cantilever(n, 1, dim)
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    u = c.sim_solve(rtol=1e-10)
    i, g, p = c.last_info, c.multigrid_info(), c.precond_info()
    assert i["converged"] and i["true_rel_residual"] < 2e-10 and p["note"] == ""
    assert g["fine_dof"] == g["coarse_dof"] == c.n_dof and g["lambda_max_fine"] == 0 and g["lambda_max_coarse"] > 1.0 and p["aggregates"] > 0
    sim = O.Simulator(T, V, 1)
    sim.set_material_constant(O.ElasticityTensor.isotropic(dim, 200.0, 0.35))
    lo, hi = [-9.0] * dim, [9.0] * dim
    lo[0], hi[0] = -1e-9, 1e-9
    sim.apply_dirichlet_box(lo, hi, [0.0] * dim)
    lo[0], hi[0] = 1 - 1e-9, 1 + 1e-9
    sim.apply_neumann_box(lo, hi, [0.0, -1.0, 0.0][:dim], "traction")
    uref = sim.solve()
    assert np.linalg.norm(u - uref) / np.linalg.norm(uref) < U_RTOL
    it_mg = i["iterations"]
    c.set_preconditioner(M.PRECOND_BLOCK_JACOBI)
    u2 = c.sim_solve(rtol=1e-10)
    assert np.linalg.norm(u2 - u) <= 1e-7 * np.linalg.norm(u) and it_mg < 0.5 * c.last_info["iterations"]
    # deep hierarchy / dense-only variants on the same context
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    for opts in ({"mg_agg_target": 6, "mg_dense_max": 8}, {"mg_agg_target": 0}):
        for k, v in opts.items():
            c.set_option(k, v)
        u3 = c.sim_solve(rtol=1e-10)
        assert c.last_info["converged"] and np.linalg.norm(u3 - u) <= 1e-7 * np.linalg.norm(u), opts
    c.close()


def test_multigrid_on_linear_elements_iteration_count_does_not_grow_with_the_mesh():
    """BASELINE configs[1] in small (linear tets, one face clamped, traction opposite)."""
    its = {}
    for n in (10, 20):
        c, _, _ = _cantilever(n, 1)
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        c.sim_solve(rtol=1e-8)
        its[n] = c.last_info["iterations"]
        c.set_preconditioner(M.PRECOND_TWO_LEVEL)
        c.sim_solve(rtol=1e-8)
        tl = c.last_info["iterations"]
        c.close()
        assert its[n] < tl, (n, its[n], tl)
    assert its[20] <= 1.3 * its[10] + 3 and its[20] <= 60, its


def test_multigrid_on_the_scalar_operators_falls_back_with_a_note():
    V, T = grid.grid_tet_mesh(4, 4, 4, [0, 0, 0], [1, 1, 1])
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.set_operator(M.OP_LAPLACIAN)
    c.bc_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9], [1.0, 0, 0], components=[True, False, False])
    c.bc_dirichlet_box([1 - 1e-9, -9, -9], [1 + 1e-9, 9, 9], [0.0, 0, 0], components=[True, False, False])
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    x = c.sim_solve(rtol=1e-10)
    assert c.last_info["converged"] and c.multigrid_info()["fine_dof"] == 0 and c.precond_info()["note"] != ""
    assert np.abs(x.ravel() - (1 - c.node_positions()[:, 0])).max() < 1e-8            # the harmonic function 1 - x
    c.close()


@pytest.mark.parametrize("i", [2, 3])
def test_reference_cantilever_bars_keep_the_mesh_independent_iteration_count(i):
    """The reference's own example meshes (examples/cantilever/gen.sh:5: `grid 5 2^i x 2^i x 2^i -t`): a slender 5 : 1 : 1 DOMAIN of cubic
    cells, clamped at x = 0 and loaded at the far end like examples/cantilever/cantilever.bc. The slenderness of the domain does not touch
    the V-cycle: the count of the unit cube (33 - 37), far below the two-level count, on both refinements."""
    import meshfem_amd as M
    from meshfem_amd import grid
    k = 2 ** i
    V, T = grid.grid_tet_mesh(5 * k, k, k, [0, 0, 0], [5, 1, 1])
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9], [0, 0, 0])
    c.bc_neumann_box([5 - 1e-9, -9, -9], [5 + 1e-9, 9, 9], [0, -10, 0], kind=M.NEUMANN_FORCE)
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    u = c.sim_solve(rtol=1e-8, maxit=500)
    it_mg, info = c.last_info["iterations"], dict(c.last_info)
    assert info["converged"] and info["true_rel_residual"] <= 2e-8 and it_mg <= 45, info
    c.set_preconditioner(M.PRECOND_TWO_LEVEL)
    u2 = c.sim_solve(rtol=1e-8, maxit=5000)
    assert c.last_info["iterations"] > 3 * it_mg
    assert np.linalg.norm(u - u2) <= 1e-6 * np.linalg.norm(u2)
    c.close()


def test_stretched_elements_converge_and_are_the_documented_weak_case():
    """Elements stretched 16 : 1 : 1 (the unit grid scaled along x): point smoothers on the quadratic and linear levels do not damp the
    low-energy modes along the long edges and P1 cannot represent them -- the V-cycle still converges to the same displacements, but needs
    hundreds of iterations (370 - 480 against 430 two-level at 24^3; stronger polynomial smoothers bring 154 at a higher cost
    per iteration; bins of the aggregate levels in the elements' proportions do not help). Pinned here so that a change of that behaviour,
    either way, is noticed."""
    import meshfem_amd as M
    from meshfem_amd import grid
    n = 12
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [16, 1, 1])
    res = {}
    for name, opts in (("default", ()), ("strong_smoothers", (("mg_steps_fine", 4), ("mg_ratio_fine", 0.02), ("mg_steps_coarse", 4), ("mg_ratio_coarse", 0.02)))):
        c = M.Context(0)
        for k, v in opts:
            c.set_option(k, v)
        c.mesh_build(T, V, 2)
        c.material_isotropic(200.0, 0.35)
        c.bc_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9], [0, 0, 0])
        c.bc_neumann_box([16 - 1e-9, -9, -9], [16 + 1e-9, 9, 9], [0, -1, 0], kind=M.NEUMANN_TRACTION)
        c.set_preconditioner(M.PRECOND_MULTIGRID)
        u = c.sim_solve(rtol=1e-8, maxit=5000)
        res[name] = (u, c.last_info["iterations"], c.last_info["converged"])
        c.close()
    assert res["default"][2] and res["strong_smoothers"][2]
    assert np.linalg.norm(res["default"][0] - res["strong_smoothers"][0]) <= 1e-6 * np.linalg.norm(res["default"][0])
    assert res["strong_smoothers"][1] < 0.7 * res["default"][1]          # the smoother is the weak part, not the coarse space
    assert 60 < res["default"][1] < 1500


@pytest.mark.parametrize("box,want", [([1, 1, 1], "multigrid"), ([4, 1, 1], "multigrid"), ([16, 1, 1], "two_level"), ([1, 12, 1], "two_level")])
def test_stretched_elements_get_the_preconditioner_chosen_for_them(box, want):
    """MFH_PRECOND_AUTO, the drivers' default: the stretch of the mesh as a whole (sqrt of the eigenvalue ratio of the edge
    covariance, reduced on the device) decides between the V-cycle and the two-level preconditioner at the measured crossover (8 : 1 : 1;
    profiles/r06_auto_preconditioner_table.jsonl: never more than 3 % behind the better of the two). The choice follows new vertices, and the
    answer is the fixed preconditioner's."""
    import meshfem_amd as M
    from meshfem_amd import grid
    n = 10
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], box)
    c = M.Context(0)
    c.mesh_build(T, V, 2)
    c.material_isotropic(200.0, 0.35)
    c.bc_dirichlet_box([-1e-9, -99, -99], [1e-9, 99, 99], [0, 0, 0])
    c.bc_neumann_box([box[0] - 1e-9, -99, -99], [box[0] + 1e-9, 99, 99], [0, -1, 0], kind=M.NEUMANN_TRACTION)
    c.set_preconditioner(M.PRECOND_AUTO)
    u = c.sim_solve(rtol=1e-9, maxit=5000)
    it_auto = c.last_info["iterations"]
    kind, is_auto, stretch = c.precond_choice()
    names = {M.PRECOND_MULTIGRID: "multigrid", M.PRECOND_TWO_LEVEL: "two_level"}
    assert is_auto and names[kind] == want, (kind, stretch)
    assert abs(stretch - max(box) / min(box)) <= 1e-6 * stretch          # a grid stretched s : 1 : 1 has stretch s
    c.set_preconditioner(kind)                                            # the same preconditioner, named: same iterations, same answer
    u2 = c.sim_solve(rtol=1e-9, maxit=5000)
    assert abs(c.last_info["iterations"] - it_auto) <= 2 and np.linalg.norm(u - u2) <= 1e-8 * np.linalg.norm(u2)      # (LDS sums in arrival order: +-1 iteration)
    if want == "two_level":
        # new vertices (the mesh squeezed back to a cube): the next solve chooses again
        c.set_preconditioner(M.PRECOND_AUTO)
        c.mesh_update_vertices(V / np.asarray(box, dtype=np.float64))
        assert names[c.precond_choice()[0]] == "multigrid" and abs(c.precond_choice()[2] - 1.0) < 1e-6
    c.close()


@pytest.mark.parametrize("deg,n", [(2, 8), (1, 14)])
def test_fp32_copies_of_the_coarse_operators_change_neither_the_answer_nor_the_iteration_count(deg, n):
    """Option mg_coarse_fp32 (default on): inside the preconditioner the linear level's assembled K and the aggregate stencils are read
    from FP32 copies, products and sums in FP64. The preconditioner stays one fixed SPD operator, so the PCG converges to the FP64
    solution of K u = f -- checked against the FP64-storage hierarchy and, through it, the oracle's direct solve -- in the same number
    of iterations (+-1). A re-assembly drops the copy (the smoother then reads the FP64 matrix) and the solve still converges."""
    c, V, T = _cantilever(n, deg)
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    res = {}
    for fp32 in (1, 0, 1):
        c.set_option("mg_coarse_fp32", fp32)
        u = c.sim_solve(rtol=1e-10)
        assert c.last_info["converged"] and c.last_info["true_rel_residual"] < 2e-10
        res[fp32] = (u, c.last_info["iterations"])
    assert abs(res[1][1] - res[0][1]) <= 1
    assert np.linalg.norm(res[1][0] - res[0][0]) <= 1e-8 * np.linalg.norm(res[0][0])
    sim = O.Simulator(T, V, deg)
    sim.set_material_constant(O.ElasticityTensor.isotropic(3, 200.0, 0.35))
    sim.apply_dirichlet_box([-1e-9, -9, -9], [1e-9, 9, 9], [0, 0, 0])
    sim.apply_neumann_box([1 - 1e-9, -9, -9], [1 + 1e-9, 9, 9], [0, -1, 0], "traction")
    uref = sim.solve()
    assert np.linalg.norm(res[1][0] - uref) <= U_RTOL * np.linalg.norm(uref)
    c.assemble()                         # rewrites the values: the FP32 copy is gone, the hierarchy stays
    u2 = c.sim_solve(rtol=1e-10)
    assert c.last_info["converged"]
    assert np.linalg.norm(u2 - res[0][0]) <= 1e-8 * np.linalg.norm(res[0][0])
    c.close()


@pytest.mark.parametrize("dim,n", [(3, 5), (2, 14)])
def test_fused_pcg_kernels_walk_the_iterates_of_the_separate_ones(dim, n):
    """Option mg_fuse (default on): r -= alpha Ap and the V-cycle's first smoothing step share a kernel, its last step and r.z another (MgFuse,
    k_pcg_update's ZS flavour, k_mg_cheb_rz). The arithmetic per entry is the same, only the order of the r.z sum differs: the same iteration
    counts and the same answer to rounding, with Dirichlet rows (masked), a deterministic run and several right-hand sides. (The kernels take two
    rows per lane; the generator's 3D grids have odd node counts, so their single-row tail runs here too.)"""
    c, V, T = _cantilever(n, 2, dim)
    c.set_preconditioner(M.PRECOND_MULTIGRID)
    # the default also reads an FP32 copy of the inverse diagonal blocks in the two fused kernels (option mg_dinv_fp32; the smoother only): another
    # preconditioner in the 8th digit -- the same answer to the tolerance, iteration counts within one
    u32 = c.sim_solve(rtol=1e-10)
    it32, res32 = c.last_info["iterations"], c.last_info["true_rel_residual"]
    c.set_option("mg_dinv_fp32", 0)
    out = {}
    for fuse in (1, 0):
        c.set_option("mg_fuse", fuse)
        u = c.sim_solve(rtol=1e-10)
        out[fuse] = (u, c.last_info["iterations"], c.last_info["true_rel_residual"])
    assert out[1][1] == out[0][1] and out[1][2] < 2e-10
    assert np.linalg.norm(out[1][0] - out[0][0]) <= 1e-11 * np.linalg.norm(out[0][0])
    assert abs(it32 - out[0][1]) <= 1 and res32 < 2e-10
    assert np.linalg.norm(u32 - out[0][0]) <= 1e-8 * np.linalg.norm(out[0][0])
    # deterministic reductions: two fused runs agree bit for bit
    c.set_option("mg_fuse", 1)
    c.set_option("deterministic", 1)
    a = c.sim_solve(rtol=1e-10)
    b = c.sim_solve(rtol=1e-10)
    assert np.array_equal(a, b)
    assert np.linalg.norm(a - out[0][0]) <= 1e-11 * np.linalg.norm(out[0][0])
    c.set_option("deterministic", 0)
    # several right-hand sides (the batched V-cycle takes the same switch)
    rng = np.random.default_rng(3)
    nrhs = 6 if dim == 3 else 3
    F = rng.standard_normal((nrhs, c.n_dof * dim))
    res = {}
    for fuse in (1, 0):
        c.set_option("mg_fuse", fuse)
        U, infos = c.solve_batch(F, rtol=1e-10)
        res[fuse] = (U, [i["iterations"] for i in infos])
    assert res[1][1] == res[0][1]
    assert np.linalg.norm(res[1][0] - res[0][0]) <= 1e-10 * np.linalg.norm(res[0][0])
    c.close()
