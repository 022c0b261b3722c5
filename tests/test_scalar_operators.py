"""Scalar operators: Laplacian.hh / MassMatrix.hh / Poisson.hh on the elasticity path's
machinery with 1x1 blocks. CPU part: the oracle restatement against closed forms; GPU part: the HIP path
against the oracle. FP64 tolerances are stated per assertion."""
import numpy as np
import pytest

from oracle import meshfem_oracle as O


def _mesh(dim, deg, seed=0):
    if dim == 3:
        V, T = O.grid_tet_mesh(3, 2, 2)
    else:
        V, Q = O.gen_grid_2d(4, 3)
        V, T = O.quad_tri_subdiv(V, Q)
        V = V[:, :2]
    rng = np.random.default_rng(seed)
    V = V + 0.08 * rng.standard_normal(V.shape)       # generic geometry, orientation preserved
    return O.FEMMesh(T, V, deg)


@pytest.mark.parametrize("dim", [2, 3])
def test_oracle_p1_closed_forms(dim):
    """Degree 1: Laplacian.hh:60-81 (grad lambda_i . grad lambda_j vol) and the textbook mass matrix
    vol (1 + delta_ij) / ((K+1)(K+2))."""
    m = _mesh(dim, 1)
    vol, gl = m.embeddings_batch()
    n = m.num_nodes
    Lref, Mref = np.zeros((n, n)), np.zeros((n, n))
    K = m.K
    for e, nodes in enumerate(m.elem_nodes):
        Lref[np.ix_(nodes, nodes)] += gl[e].T @ gl[e] * vol[e]
        Mref[np.ix_(nodes, nodes)] += vol[e] * (np.ones((K + 1, K + 1)) + np.eye(K + 1)) / ((K + 1) * (K + 2))
    Lo = O.laplacian_triplets(m).sum_repeated().to_scipy_full_from_upper().toarray()
    Mo = O.mass_triplets(m).sum_repeated().to_scipy_full_from_upper().toarray()
    assert np.abs(Lo - Lref).max() < 1e-13 * np.abs(Lref).max()
    assert np.abs(Mo - Mref).max() < 1e-14 * np.abs(Mref).max()


@pytest.mark.parametrize("dim,deg", [(2, 1), (2, 2), (3, 1), (3, 2)])
def test_oracle_operator_identities(dim, deg):
    m = _mesh(dim, deg)
    vol, _ = m.embeddings_batch()
    L = O.laplacian_triplets(m).sum_repeated().to_scipy_full_from_upper()
    M = O.mass_triplets(m).sum_repeated().to_scipy_full_from_upper()
    one = np.ones(m.num_nodes)
    assert np.abs(L @ one).max() < 1e-12                                  # constants are in the kernel
    assert abs(one @ (M @ one) - vol.sum()) < 1e-12 * vol.sum()           # int 1 = volume
    x = m.node_pos
    # u^T L u = int |grad u|^2 and u^T M u = int u^2 for (piecewise-)polynomial u in the FE space
    a = np.arange(1, dim + 1, dtype=np.float64)
    u = x @ a
    assert abs(u @ (L @ u) - (a @ a) * vol.sum()) < 1e-11 * vol.sum() * (a @ a)
    Ml = O.mass_triplets(m, lumped=True)
    assert np.array_equal(Ml.i, Ml.j) and abs(Ml.v.sum() - vol.sum()) < 1e-12 * vol.sum()
    assert np.abs(Ml.v - M @ one).max() < 1e-14 * np.abs(Ml.v).max()


@pytest.mark.parametrize("dim,deg", [(2, 2), (3, 1), (3, 2)])
def test_oracle_poisson_patch(dim, deg):
    if dim == 3:
        V, T = O.grid_tet_mesh(3, 2, 2)
    else:
        V, Q = O.gen_grid_2d(4, 3)
        V, T = O.quad_tri_subdiv(V, Q)
        V = V[:, :2]
    m = O.FEMMesh(T, V, deg)
    big = 1e9
    xmax = V[:, 0].max()
    u, fixed = O.poisson_solve(m, [([-1e-9] + [-big] * (dim - 1), [1e-9] + [big] * (dim - 1), 0.0),
                                   ([xmax - 1e-9] + [-big] * (dim - 1), [xmax + 1e-9] + [big] * (dim - 1), 2.0 * xmax)])
    assert len(fixed) > 0
    assert np.abs(u - 2.0 * m.node_pos[:, 0]).max() < 1e-12 * xmax
    g = O.grad_u_average(m, u)
    assert np.abs(g - np.eye(dim)[0] * 2.0).max() < 1e-12


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dim,deg", [(2, 1), (2, 2), (3, 1), (3, 2)])
def test_gpu_scalar_operators_match_oracle(dim, deg):
    import meshfem_amd as M
    from meshfem_amd import scalar_operators as S
    m = _mesh(dim, deg, seed=3)
    c = M.Context(0)
    c.mesh_build(m.elems, m.verts, deg)
    assert np.array_equal(c.elem_nodes(), m.elem_nodes)
    n = m.num_nodes
    vol, gl = m.embeddings_batch()
    for op, trip in ((M.OP_LAPLACIAN, O.laplacian_triplets(m)), (M.OP_MASS, O.mass_triplets(m))):
        c.set_operator(op)
        ref = trip.sum_repeated()
        A_ref = ref.to_scipy_full_from_upper().toarray()
        # element matrices (npe x npe)
        Ke = c.element_stiffness()
        assert Ke.shape == (len(m.elems), m.nodes_per_elem, m.nodes_per_elem)
        acc = np.zeros((n, n))
        for e, nodes in enumerate(m.elem_nodes):
            acc[np.ix_(nodes, nodes)] += Ke[e]
        assert np.abs(acc - A_ref).max() < 1e-13 * np.abs(A_ref).max()
        # assembled matrix, both assembly variants; exported upper triplets in dumpBinary order
        for mode in (M.ASSEMBLE_GATHER, M.ASSEMBLE_ATOMIC):
            c.assemble(mode)
            A = c.export_scipy().toarray()
            assert A.shape == (n, n)
            assert np.abs(A - A_ref).max() < 1e-13 * np.abs(A_ref).max()
        i, j, v = c.export_upper_triplets()
        assert (i <= j).all() and np.all(np.diff(j.astype(np.int64)) >= 0)
        U = np.zeros((n, n)); U[i.astype(int), j.astype(int)] = v
        assert np.abs(U - np.triu(A_ref)).max() < 1e-13 * np.abs(A_ref).max()
        # SpMV
        x = np.random.default_rng(1).standard_normal(n)
        assert np.abs(c.apply_K(x) - A_ref @ x).max() < 1e-12 * np.abs(A_ref @ x).max()
        c.set_option("matrix_free", 1)                      # per-pair matrix-free variant (k_spmv_mf) on 1x1 blocks
        assert np.abs(c.apply_K(x) - A_ref @ x).max() < 1e-12 * np.abs(A_ref @ x).max()
        c.set_option("matrix_free", -1)
    # the elasticity operator still works on the same context afterwards (shared pattern)
    c.set_operator(M.OP_ELASTICITY)
    c.material_isotropic(200.0, 0.35)
    sim = O.Simulator(m.elems, m.verts, deg)
    sim.set_material_constant(O.ElasticityTensor.isotropic(dim, 200.0, 0.35))
    K_ref = sim.assembleStiffnessMatrix().sum_repeated().to_scipy_full_from_upper().toarray()
    c.assemble()
    assert np.abs(c.export_scipy().toarray() - K_ref).max() < 1e-13 * np.abs(K_ref).max()
    # mirrors
    Lt = S.laplacian(None, None, ctx=c)
    assert np.abs(Lt.toSciPy().toarray() - O.laplacian_triplets(m).sum_repeated().to_scipy_full_from_upper().toarray()).max() < 1e-12
    Ml = S.mass_matrix(None, None, lumped=True, ctx=c)
    assert np.abs(Ml.v - O.mass_triplets(m, lumped=True).v).max() < 1e-14 * np.abs(Ml.v).max()


@pytest.mark.gpu
@pytest.mark.parametrize("dim,deg", [(2, 2), (3, 1), (3, 2)])
def test_gpu_poisson_matches_oracle(dim, deg):
    from meshfem_amd import scalar_operators as S
    m = _mesh(dim, deg, seed=5)
    x = m.node_pos
    lo, hi = x.min(axis=0), x.max(axis=0)
    span = hi - lo
    # two overlapping regions: the later condition overwrites the earlier one on shared boundary nodes
    boxes = [(lo - 1e-9, np.concatenate([[lo[0] + 0.3 * span[0]], hi[1:] + 1e-9]), 1.0),
             (np.concatenate([[hi[0] - 0.3 * span[0]], lo[1:] - 1e-9]), hi + 1e-9, -2.0),
             (lo - 1e-9, np.concatenate([[lo[0] + 0.15 * span[0]], hi[1:] + 1e-9]), 0.5)]
    u_ref, fixed = O.poisson_solve(m, boxes)
    pm = S.PoissonMesh(m.elems, m.verts, deg)
    for mn, mx, val in boxes:
        pm.applyDirichletBox(mn, mx, val)
    fv, fx = pm.ctx.bc_dirichlet_vars()
    assert np.array_equal(fv, fixed)
    u = pm.solve()
    assert pm.info["converged"] == 1
    # tolerance: PCG to rtol 1e-10 against a direct solve
    assert np.linalg.norm(u - u_ref) < 1e-8 * np.linalg.norm(u_ref)
    g = pm.gradUAverage(u_ref)
    assert np.abs(g - O.grad_u_average(m, u_ref)).max() < 1e-12 * max(1.0, np.abs(g).max())
    # two-level request falls back to Jacobi with a note instead of failing
    import meshfem_amd as M
    pm.ctx.set_preconditioner(M.PRECOND_TWO_LEVEL)
    u2 = pm.solve()
    assert "elasticity only" in pm.ctx.precond_info()["note"]
    assert np.linalg.norm(u2 - u_ref) < 1e-8 * np.linalg.norm(u_ref)


@pytest.mark.gpu
def test_gpu_scalar_operators_at_scale_properties():
    """35^3 grid (1.03 M tets, P2: 1.4 M nodes): size-independent properties -- L 1 = 0, 1^T M 1 = volume,
    u^T L u = |a|^2 volume for u = a.x, Poisson patch test."""
    import meshfem_amd as M
    from meshfem_amd import grid, scalar_operators as S
    n = 35
    V, T = grid.grid_tet_mesh(n, n, n, [0, 0, 0], [1, 1, 1])
    pm = S.PoissonMesh(T, V, 2)
    c = pm.ctx
    N = c.n_node
    one = np.ones(N)
    x = c.node_positions()
    a = np.array([1.0, -2.0, 0.5])
    u = x @ a
    assert np.abs(c.apply_K(one)).max() < 1e-11
    assert abs(u @ c.apply_K(u) - a @ a) < 1e-10 * (a @ a)
    pm.applyDirichletBox([-1e-9, -9, -9], [1e-9, 9, 9], 0.0)
    pm.applyDirichletBox([1 - 1e-9, -9, -9], [1 + 1e-9, 9, 9], 3.0)
    pm.rtol = 1e-12
    uh = pm.solve()
    assert np.abs(uh - 3.0 * x[:, 0]).max() < 1e-8
    c.set_operator(M.OP_MASS)
    assert abs(one @ c.apply_K(one) - 1.0) < 1e-12
    assert abs(u @ c.apply_K(u) - sum(a[i] * a[j] * (1 / 3 if i == j else 1 / 4) for i in range(3) for j in range(3))) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("dim,deg", [(2, 1), (2, 2), (3, 1), (3, 2)])
def test_gpu_mass_matrix_l2_norm_validation(dim, deg):
    """tests/test_mass.cc:6-45 on the HIP path: u^T M u (device SpMV with the assembled mass matrix) against the
    direct quadrature of |u_h|^2 on the reference's meshes; 16 random fields, 1e-13 relative."""
    import os
    import meshfem_amd as M
    from meshfem_amd import mesh_io
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "meshes")
    V, E, _ = mesh_io.load_mesh(os.path.join(gold, "square_hole.off" if dim == 2 else "ball.msh"))
    m = O.FEMMesh(E, V[:, :dim], deg)
    c = M.Context(0)
    c.mesh_build(m.elems, m.verts, deg)
    c.set_operator(M.OP_MASS)
    vol, _ = m.embeddings_batch()
    pts, w = O.quadrature_rule(m.K, 2 * deg)
    Phi = np.array([O.shape_functions(deg, m.K, p) for p in pts])
    rng = np.random.default_rng(0)
    for _ in range(16):
        u = rng.uniform(-1, 1, (m.num_nodes, dim))
        l2_mass = sum(u[:, k] @ c.apply_K(u[:, k].copy()) for k in range(dim))
        uq = np.einsum("qn,enc->eqc", Phi, u[m.elem_nodes])
        direct = float(np.einsum("q,e,eqc,eqc->", w, vol, uq, uq))
        assert abs(l2_mass - direct) < 1e-13 * abs(l2_mass)
