"""CPU tests pinning the oracle (oracle/) against the golden vectors the reference's own tests
hold (ported from /root/reference/tests/*.cc, cited per test) and against exact mathematics."""
import json
import os
from fractions import Fraction

import numpy as np
import pytest

from oracle import meshfem_oracle as O
from oracle import c_oracle as CO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- tests/test_quadrature.cc:26-40,137-170: every rule integrates the monomial basis up to its degree
#      on the unit-volume simplex with relative error <= 1e-15
def test_quadrature_monomials_reference_tables():
    with open(os.path.join(GOLD, "quadrature_monomials.json")) as f:
        tables = json.load(f)
    maxdeg = {1: 4, 2: 5, 3: 4}
    for K in (1, 2, 3):
        for rule_deg in range(0, maxdeg[K] + 1):
            for d in range(0, rule_deg + 1):
                for entry in tables[str(K)][d]:
                    ex = entry["exponents"]
                    exact = float(Fraction(entry["value"]))
                    val = O.integrate(K, rule_deg, lambda p: np.prod([p[k] ** ex[k] for k in range(len(ex))]), 1.0)
                    assert abs((val - exact) / exact) <= 1e-15, (K, rule_deg, ex, val, exact)


# ---- tests/test_shape_functions.cc:14-66: gradPhi(j)(x) == gradPhis(x).col(j); integrated phis
@pytest.mark.parametrize("K,deg", [(2, 1), (2, 2), (3, 1), (3, 2)])
def test_shape_function_gradients_self_consistent(K, deg):
    rng = np.random.default_rng(K * 10 + deg)
    n = O.num_nodes(K, deg)
    for _ in range(100):
        P = rng.standard_normal((K + 1, K))
        vol, gl = O.embed(K, P)
        for _ in range(20):
            x = rng.random(K + 1)
            x /= x.sum()
            G = O.grad_phis_at(deg, K, gl, x)
            for j in range(n):
                gj = O.eval_interpolant(K, deg - 1, O.grad_phi_nodal(deg, K, gl, j), x)
                assert np.abs(gj - G[:, j]).max() <= 1e-12 * max(1.0, np.abs(G).max())
        # phis sum to one; integral of phi via quadrature equals integratedPhis
        x = rng.random(K + 1); x /= x.sum()
        assert abs(O.shape_functions(deg, K, x).sum() - 1) < 1e-14
        quad = O.integrate(K, deg, lambda p: O.shape_functions(deg, K, p), 1.0)
        assert np.abs(quad - O.integrated_shape_functions(deg, K)).max() < 1e-15
        # partition of unity of the gradients
        assert np.abs(G.sum(axis=1)).max() < 1e-10 * np.abs(G).max()


# ---- tests/test_tensors.cc:4-27: flatten/unflatten round trip
def test_flatten_unflatten_roundtrip():
    for dim in (2, 3):
        for k in range(O.flat_len(dim)):
            i, j = O.unflatten_index(dim, k)
            assert O.flatten_indices(dim, i, j) == k and O.flatten_indices(dim, j, i) == k
    assert [O.flatten_indices(3, *p) for p in [(0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1)]] == [0, 1, 2, 3, 4, 5]
    assert [O.flatten_indices(2, *p) for p in [(0, 0), (1, 1), (0, 1)]] == [0, 1, 2]


# ---- tests/test_sparse_matrices.cc:7-160: triplet <-> CSC round trip, symmetric upper apply
def test_triplet_csc_roundtrip_and_symmetric_apply():
    rng = np.random.default_rng(0)
    n = 40
    dense = rng.standard_normal((n, n)) * (rng.random((n, n)) < 0.15)
    dense = dense + dense.T
    iu, ju = np.nonzero(np.triu(dense))
    # duplicate-split every entry to exercise sumRepeated
    i = np.concatenate([iu, iu]); j = np.concatenate([ju, ju])
    v = np.concatenate([0.25 * dense[iu, ju], 0.75 * dense[iu, ju]])
    T = O.TripletMatrix.from_arrays(n, n, i, j, v)
    Ap, Ai, Ax = T.to_csc()
    assert len(Ax) == len(iu)
    x = rng.standard_normal(n)
    y = O.csc_apply_symmetric_upper(Ap, Ai, Ax, x)
    assert np.abs(y - dense @ x).max() <= 1e-14 * np.abs(dense @ x).max() + 1e-15
    full = T.to_scipy_full_from_upper().toarray()
    assert np.abs(full - dense).max() < 5e-16 * np.abs(dense).max() + 1e-16
    # exact zeros are pruned (pruneTol = 0)
    Z = O.TripletMatrix.from_arrays(3, 3, [0, 0, 1], [1, 1, 2], [1.0, -1.0, 2.0]).sum_repeated()
    assert Z.nnz() == 1 and Z.i[0] == 1 and Z.j[0] == 2


# ---- exact (sympy rational) element stiffness: anchors A6 independent of any quadrature rule
def test_ke_matches_exact_rational_integration():
    with open(os.path.join(GOLD, "ke_exact.json")) as f:
        cases = json.load(f)
    assert len(cases) == 16
    for cs in cases:
        K, deg = cs["K"], cs["deg"]
        P = np.array(cs["verts"])
        vol, gl = O.embed(K, P)
        assert abs(vol - cs["vol"]) < 1e-14
        ten = O.ElasticityTensor(K, np.array(cs["D"]))
        exact = np.array(cs["Ke"])
        loop = O.per_element_stiffness_loop(deg, K, gl, vol, ten)
        batch = O.per_element_stiffness_batch(deg, K, gl[None], np.array([vol]), ten.rank4()[None])[0]
        iu = np.triu_indices(exact.shape[0])
        s = np.abs(exact).max()
        assert np.abs(loop[iu] - exact[iu]).max() < 5e-14 * s
        assert np.abs(batch - exact).max() < 5e-14 * s
        if K == 3:
            en = np.arange(4)[None, :]
            Kc, _ = CO.element_stiffness(3, deg, _p2_nodes(deg), _p2_verts(P, deg), ten.D)
            assert np.abs(Kc[0][iu] - exact[iu]).max() < 5e-14 * s


def _p2_nodes(deg):
    return np.arange(4 if deg == 1 else 10, dtype=np.int32)[None, :]


def _p2_verts(P, deg):
    return P        # only the 4 corner nodes are read by the embedding


# ---- patch properties of Ke (rigid modes, PSD, constant-strain energy)
@pytest.mark.parametrize("deg", [1, 2])
def test_ke_properties(deg):
    rng = np.random.default_rng(7)
    ten = O.ElasticityTensor.orthotropic3d(150, 200, 250, 0.3, 0.25, 0.2, 60, 70, 80)
    for _ in range(5):
        P = rng.random((4, 3))
        vol, gl = O.embed_tet(P)
        if vol < 0:
            P[[0, 1]] = P[[1, 0]]
            vol, gl = O.embed_tet(P)
        Ke = O.per_element_stiffness_batch(deg, 3, gl[None], np.array([vol]), ten.rank4()[None])[0]
        w = np.linalg.eigvalsh(Ke)
        assert np.sum(np.abs(w) < 1e-9 * w.max()) == 6 and w.min() > -1e-9 * w.max()
        # nodes: corners + edge midpoints; linear field u = G x  =>  u^T K u = vol * eps:C:eps
        nodes = P if deg == 1 else np.vstack([P, [0.5 * (P[O.EDGE_START[e]] + P[O.EDGE_END[e]]) for e in range(6)]])
        G = rng.standard_normal((3, 3))
        u = (nodes @ G.T).ravel()
        eps = 0.5 * (G + G.T)
        energy = vol * np.sum(eps * ten.double_contract(eps))
        assert abs(u @ Ke @ u - energy) < 1e-11 * abs(energy)


# ---- committed end-to-end fixture of the oracle (see tests/golden/make_goldens.py)
@pytest.mark.parametrize("deg", [1, 2])
def test_cantilever_fixture_reproduced(deg):
    g = np.load(os.path.join(GOLD, "cantilever_small.npz"))
    sim = O.Simulator(g["T"], g["V"], deg)
    assert np.array_equal(sim.mesh.elem_nodes, g["p%d_elem_nodes" % deg])
    assert np.array_equal(sim.mesh.bdry_elem_nodes, g["p%d_bdry_elem_nodes" % deg])
    assert np.array_equal(sim.mesh.bdry_nodes, g["p%d_bdry_nodes" % deg])
    sim.set_material_constant(O.ElasticityTensor.isotropic(3, 200.0, 0.35))
    mn, mx = sim.box_percent([-1e-4] * 3, [1e-4, 1.0001, 1.0001]); sim.apply_dirichlet_box(mn, mx, [0, 0, 0])
    mn, mx = sim.box_percent([0.9999, -1e-4, -1e-4], [1.0001, 1.0001, 1.0001]); sim.apply_neumann_box(mn, mx, [0, -10, 0], "force")
    Kt = sim.assembleStiffnessMatrix().sum_repeated()
    import scipy.sparse as sp
    n = Kt.m
    A = sp.coo_matrix((Kt.v, (Kt.i, Kt.j)), shape=(n, n)).tocsr()
    B = sp.coo_matrix((g["p%d_K_v" % deg], (g["p%d_K_i" % deg], g["p%d_K_j" % deg])), shape=(n, n)).tocsr()
    assert abs(A - B).max() < 1e-12 * abs(B).max()
    assert np.abs(sim.neumannLoad() - g["p%d_load" % deg]).max() < 1e-13
    fv, _ = sim.dirichlet_vars_and_values()
    assert np.array_equal(np.array(fv), g["p%d_fixed_vars" % deg])
    u = sim.solve()
    assert np.linalg.norm(u - g["p%d_u" % deg]) < 1e-9 * np.linalg.norm(u)
    # the C restatement assembles the same matrix
    Ap, Ai, Ax, _ = CO.assemble_csc(3, deg, sim.mesh.elem_nodes, g["V"], sim.D[0].D, sim.mesh.num_nodes)
    C = sp.csc_matrix((Ax, Ai, Ap), shape=(n, n)).tocsr()
    assert abs(C - B).max() < 1e-12 * abs(B).max()
    # ... and so does the restructured host assembly bench.py times beside it (cpu_baseline.tuned)
    Ax2, _, _ = CO.assemble_fused(3, deg, sim.mesh.elem_nodes, g["V"], sim.D[0].D, sim.mesh.num_nodes, Ap, Ai)
    assert np.abs(Ax2 - Ax).max() < 1e-13 * np.abs(Ax).max()


EXAMPLE_BCS = {"cube_cross": (([-1e-3] * 3, [0.02, 1.001, 1.001]), ([0.98, -1e-3, -1e-3], [1.001] * 3), [0, -1, 0]),
               "ball": (([-1e-3] * 3, [1.001, 1.001, 0.12]), ([-1e-3, -1e-3, 0.88], [1.001] * 3), [0.3, 0, -1])}


@pytest.mark.parametrize("name,deg", [("cube_cross", 1), ("cube_cross", 2), ("ball", 1)])
def test_example_mesh_fixture_reproduced(name, deg):
    """Unstructured example meshes of the reference (tests/golden/meshes, data files): the oracle reproduces
    its committed direct-solve displacements, load, Dirichlet variables and matrix checksums."""
    from meshfem_amd import mesh_io
    g = np.load(os.path.join(GOLD, "example_meshes.npz"))
    V, E, _ = mesh_io.load_msh(os.path.join(GOLD, "meshes", name + ".msh"))
    lo_box, hi_box, trac = EXAMPLE_BCS[name]
    sim = O.Simulator(E, V, deg)
    sim.set_material_constant(O.ElasticityTensor.isotropic(3, 200.0, 0.35))
    mn, mx = sim.box_percent(*lo_box); sim.apply_dirichlet_box(mn, mx, [0, 0, 0])
    mn, mx = sim.box_percent(*hi_box); sim.apply_neumann_box(mn, mx, trac, "traction")
    key = "%s_p%d_" % (name, deg)
    fv, _ = sim.dirichlet_vars_and_values()
    assert np.array_equal(np.array(fv), g[key + "fixed_vars"])
    assert np.abs(sim.neumannLoad() - g[key + "load"]).max() < 1e-13
    Kt = sim.assembleStiffnessMatrix().sum_repeated()
    assert Kt.nnz() == g[key + "K_nnz"][0]
    chk = np.array([Kt.v.sum(), np.abs(Kt.v).sum(), (Kt.v * (1 + Kt.i % 7) * (1 + Kt.j % 5)).sum()])
    assert np.abs(chk - g[key + "K_checksum"]).max() < 1e-10 * np.abs(g[key + "K_checksum"]).max()
    u = sim.solve()
    assert np.linalg.norm(u - g[key + "u"]) < 1e-9 * np.linalg.norm(u)


@pytest.mark.parametrize("name,dim,deg", [("cube_cross", 3, 1), ("2D_microstructure", 2, 1), ("2D_microstructure", 2, 2)])
def test_example_mesh_homogenization_fixture_reproduced(name, dim, deg):
    from meshfem_amd import mesh_io
    g = np.load(os.path.join(GOLD, "example_meshes.npz"))
    V, E, _ = mesh_io.load_msh(os.path.join(GOLD, "meshes", name + ".msh"))
    sim = O.Simulator(E, V[:, :dim], deg)
    base = O.ElasticityTensor.isotropic(dim, 200.0, 0.35)
    sim.set_material_constant(base)
    w = O.solve_cell_problems(sim)
    Ch = O.homogenized_elasticity_tensor(sim, w)
    key = "%s_hom_p%d_" % (name, deg)
    assert sim.numDoFs() == g[key + "ndof"][0]
    assert np.abs(Ch - g[key + "Ch"]).max() < 1e-9 * np.abs(Ch).max()
    assert np.abs(np.array(w) - g[key + "w"]).max() < 1e-8 * np.abs(g[key + "w"]).max()
    # physics the fixture must obey: major symmetry, positive definite, softer than the base material
    assert np.abs(Ch - Ch.T).max() < 1e-9 * np.abs(Ch).max()
    assert np.linalg.eigvalsh(Ch).min() > 0
    assert np.linalg.eigvalsh(base.D - Ch).min() > 0


def test_isotropic_and_orthotropic_tensors():
    # ElasticityTensor.hh:100-134: Lame parameters; 2D is plane stress
    t = O.ElasticityTensor.isotropic(3, 200.0, 0.35)
    lam, mu = 0.35 * 200 / (1.35 * 0.3), 200 / 2.7
    assert abs(t.D[0, 0] - (lam + 2 * mu)) < 1e-12 and abs(t.D[0, 1] - lam) < 1e-12 and abs(t.D[3, 3] - mu) < 1e-12
    t2 = O.ElasticityTensor.isotropic(2, 200.0, 0.35)
    assert abs(t2.D[0, 1] - 0.35 * 200 / (1 - 0.35 ** 2)) < 1e-12
    # an isotropic material expressed through the orthotropic constructor gives the same D
    E, nu = 200.0, 0.35
    o = O.ElasticityTensor.orthotropic3d(E, E, E, nu, nu, nu, mu, mu, mu)
    assert np.abs(o.D - t.D).max() < 1e-10
    # double contraction doubles the shear entries (ElasticityTensor.hh:437-449)
    eps = np.array([[1.0, 0.2, 0.3], [0.2, -0.5, 0.1], [0.3, 0.1, 0.7]])
    sig = t.double_contract(eps)
    assert np.abs(sig - (lam * np.trace(eps) * np.eye(3) + 2 * mu * eps)).max() < 1e-12


def test_homogenization_of_solid_cube_returns_base_material():
    """a homogeneous periodic cell homogenizes to its own tensor; fluctuations vanish."""
    V, T = O.grid_tet_mesh(2, 2, 2)
    sim = O.Simulator(T, V, 1)
    base = O.ElasticityTensor.orthotropic3d(150, 200, 250, 0.3, 0.25, 0.2, 60, 70, 80)
    sim.set_material_constant(base)
    w = O.solve_cell_problems(sim)
    assert max(np.abs(x).max() for x in w) < 1e-10
    Ch = O.homogenized_elasticity_tensor(sim, w)
    # Ch_ij reported with engineering-free (tensor) strains: column j = C : e_j
    ref = np.column_stack([base.double_contract_flat(O.flatten_sym(3, O.canonical_strain(3, j))) for j in range(6)])
    assert np.abs(Ch - ref).max() < 1e-9 * np.abs(ref).max()


# ---- tests/test_mass.cc:6-45 "L2 Norm Validation": u^T M u == quadrature of |u_h|^2 with Quadrature<N, 2 Deg>,
# on the reference's own meshes (square_hole.off in 2D, ball.msh in 3D), tolerance 1e-14 relative like the reference
def _mass_l2_case(dim, deg):
    from meshfem_amd import mesh_io
    path = os.path.join(GOLD, "meshes", "square_hole.off" if dim == 2 else "ball.msh")
    V, E, _ = mesh_io.load_mesh(path)
    return O.FEMMesh(E, V[:, :dim], deg)


def _l2sq_direct(m, u):
    vol, _ = m.embeddings_batch()
    pts, w = O.quadrature_rule(m.K, 2 * m.deg)
    Phi = np.array([O.shape_functions(m.deg, m.K, p) for p in pts])            # nq x n
    uq = np.einsum("qn,enc->eqc", Phi, u[m.elem_nodes])                          # u_interp(p) at every point
    return float(np.einsum("q,e,eqc,eqc->", w, vol, uq, uq))


@pytest.mark.parametrize("dim,deg", [(2, 1), (2, 2), (3, 1), (3, 2)])
def test_mass_matrix_l2_norm_validation(dim, deg):
    m = _mass_l2_case(dim, deg)
    Ms = O.mass_triplets(m).sum_repeated().to_scipy_full_from_upper()           # construct_vector_valued = scalar M per component
    rng = np.random.default_rng(0)
    for _ in range(16):
        u = rng.uniform(-1, 1, (m.num_nodes, dim))                              # Eigen setRandom: uniform in [-1, 1]
        l2_mass = sum(u[:, c] @ (Ms @ u[:, c]) for c in range(dim))
        assert abs(l2_mass - _l2sq_direct(m, u)) < 1e-13 * abs(l2_mass)


# ---- tests/test_materials.cc:28-90: the reference's sample material files survive read -> getJson -> read -> getJson
MATERIAL_SAMPLES = {
    "isotropic": ({"type": "isotropic", "young": 200, "poisson": 0.3},) * 2,
    "orthotropic": ({"type": "orthotropic", "young": [2.933545, 2.933545], "poisson": [0.27186, 0.27186], "shear": [0.87212]},
                    {"type": "orthotropic", "young": [1.0, 2.0, 3.0], "poisson": [0.6, 0.9, 0.9, 0.3, 0.3, 0.6], "shear": [0.1, 0.2, 0.3]}),
    "anisotropic": ({"type": "anisotropic", "material_matrix": [[9.0, 0.1, 0.2], [0.1, 9.0, 0.3], [0.2, 0.3, 1.0]]},
                    {"type": "anisotropic", "material_matrix": [[9.0, 0.1, 0.2, 0.5, 0.5, 0.5], [0.1, 9.0, 0.3, 0.5, 0.5, 0.5],
                                                                [0.2, 0.3, 9.0, 0.5, 0.5, 0.5], [0.5, 0.5, 0.5, 1.5, 0.1, 0.2],
                                                                [0.5, 0.5, 0.5, 0.1, 1.6, 0.3], [0.5, 0.5, 0.5, 0.2, 0.3, 1.7]]}),
}


@pytest.mark.parametrize("kind", sorted(MATERIAL_SAMPLES))
def test_material_json_roundtrip_reference_samples(kind):
    from meshfem_amd import simulate_cli
    for dim, cfg in zip((2, 3), MATERIAL_SAMPLES[kind]):
        mat = simulate_cli.parse_material(cfg, dim)
        out1 = simulate_cli.material_json(mat)
        out2 = simulate_cli.material_json(simulate_cli.parse_material(out1, dim))
        assert out1 == out2 and out1["type"] == "anisotropic"
        D = np.array(out1["material_matrix"])
        assert D.shape == ((3, 3) if dim == 2 else (6, 6)) and np.abs(D - D.T).max() < 1e-12 * np.abs(D).max()
        if kind == "isotropic":
            assert np.abs(D - O.ElasticityTensor.isotropic(dim, 200.0, 0.3).D).max() < 1e-12
        if kind == "orthotropic" and dim == 2:
            assert np.abs(D - O.ElasticityTensor.orthotropic2d(2.933545, 2.933545, 0.27186, 0.87212).D).max() < 1e-12


# ---- tests/test_interpolant.cc:28-66: a degree-Deg interpolant reproduces every monomial of degree <= Deg on all of
# R^K (20000 random points in the reference, 2000 here), and Interpolant::integrate equals Gauss quadrature
@pytest.mark.parametrize("K", [1, 2, 3])
@pytest.mark.parametrize("deg", [1, 2])
def test_interpolant_reproduces_polynomials(K, deg):
    import itertools
    rng = np.random.default_rng(0)
    # nodal positions in barycentric coordinates: vertices, then edge midpoints (Simplex.hh:30-47)
    nodes = [np.eye(K + 1)[v] for v in range(K + 1)]
    if deg == 2:
        nodes += [0.5 * (np.eye(K + 1)[O.EDGE_START[e]] + np.eye(K + 1)[O.EDGE_END[e]]) for e in range(O.num_edges(K))]
    for d in range(deg + 1):
        for expo in itertools.product(range(d + 1), repeat=K):
            if sum(expo) != d:
                continue
            f = lambda lam: float(np.prod([lam[c] ** expo[c] for c in range(K)]))      # monomial in the first K barycentric coords
            nodal = np.array([f(x) for x in nodes])
            for _ in range(2000):
                lam = np.empty(K + 1)
                lam[:K] = rng.random(K)                                             # not restricted to the simplex
                lam[K] = 1.0 - lam[:K].sum()
                assert abs(O.eval_interpolant(K, deg, nodal, lam) - f(lam)) <= 1e-13
            assert abs(O.interpolant_integrate(K, deg, nodal, 1.0) - O.integrate(K, deg, f, 1.0)) <= 1e-15
