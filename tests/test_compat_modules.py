"""The reference's Python module names (mesh, tensors, sparse_matrices, periodic_homogenization) as shims over
the C ABI (meshfem_amd/compat). CPU part: host-only behaviour; GPU part: solves against scipy / the goldens."""
import os
import sys

import numpy as np
import pytest

import meshfem_amd.compat as compat

sys.path.insert(0, compat.PATH)
import mesh as cmesh                     # noqa: E402
import tensors as ctensors               # noqa: E402
import sparse_matrices as csm            # noqa: E402
import periodic_homogenization as cph    # noqa: E402

from oracle import meshfem_oracle as O   # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_tensors_module():
    t = ctensors.ElasticityTensor3D(200.0, 0.35)
    ref = O.ElasticityTensor.isotropic(3, 200.0, 0.35)
    assert np.abs(t.D - ref.D).max() < 1e-12
    for idx in ((0, 0, 0, 0), (0, 1, 0, 1), (0, 0, 1, 1), (1, 2, 2, 1), (0, 2, 1, 1)):
        assert abs(t(*idx) - ref(*idx)) < 1e-12
    with pytest.raises(RuntimeError):
        t(3, 0, 0, 0)
    t.setOrthotropic(150, 200, 250, 0.3, 0.25, 0.2, 60, 70, 80)
    assert np.abs(t.D - O.ElasticityTensor.orthotropic3d(150, 200, 250, 0.3, 0.25, 0.2, 60, 70, 80).D).max() < 1e-10
    # inverse: S : (C : e) = e for a symmetric strain
    e = np.array([0.3, -0.1, 0.2, 0.05, -0.07, 0.11])
    assert np.abs(t.inverse().doubleContract(t.doubleContract(e)) - e).max() < 1e-13
    t2 = ctensors.ElasticityTensor2D(10.0, 0.3)
    assert abs(t2(0, 0, 1, 1) - 0.3 * 10 / (1 - 0.09)) < 1e-13
    sm = ctensors.SymmetricMatrix([1.0, 2.0, 3.0, 0.4, 0.5, 0.6])
    assert sm(1, 2) == 0.4 and sm(0, 2) == 0.5 and sm(0, 1) == 0.6 and np.allclose(sm.toMatrix(), sm.toMatrix().T)
    assert np.allclose(ctensors.SymmetricMatrix(sm.toMatrix()).flat, sm.flat)
    ident = ctensors.ElasticityTensor3D().setIdentity()
    assert np.allclose(ident.doubleContract(e), e)


def test_triplet_matrix_module(tmp_path):
    A = csm.TripletMatrix(4, 4)
    for i, j, v in ((0, 0, 2.0), (0, 1, -1.0), (1, 1, 2.0), (0, 1, 0.5), (2, 3, 1.0), (2, 3, -1.0), (3, 3, 4.0), (2, 2, 3.0)):
        A.addNZ(i, j, v)
    with pytest.raises(RuntimeError):
        A.addNZ(4, 0, 1.0)
    assert A.nnz == 8
    A.sumRepeated()                        # duplicates summed, exact zero (2,3) dropped, (col,row) order
    ents = [(t.i, t.j, t.v) for t in A.entries()]
    assert ents == [(0, 0, 2.0), (0, 1, -0.5), (1, 1, 2.0), (2, 2, 3.0), (3, 3, 4.0)]
    A.symmetry_mode = "UPPER_TRIANGLE"
    x = np.array([1.0, 2.0, 3.0, 4.0])
    full = np.array([[2, -0.5, 0, 0], [-0.5, 2, 0, 0], [0, 0, 3, 0], [0, 0, 0, 4.0]])
    assert np.allclose(A.apply(x), full @ x)
    assert np.allclose(A.diag(), np.diag(full))
    p = str(tmp_path / "a.bin")
    A.dumpBinary(p)
    B = csm.TripletMatrix()
    B.readBinary(p)
    assert (B.m, B.n, B.nnz) == (4, 4, 5) and [(t.i, t.j, t.v) for t in B.entries()] == ents
    # same bytes as the oracle's dumpBinary (SparseMatrices.hh:623-645)
    T = O.TripletMatrix.from_arrays(4, 4, *A.arrays())
    q = str(tmp_path / "b.bin")
    T.dump_binary(q)
    assert open(p, "rb").read() == open(q, "rb").read()
    A.reflectUpperTriangle()
    assert np.allclose(A.toSciPy().toarray(), full)


def test_mesh_module_host_only():
    m = cmesh.Mesh(os.path.join(GOLD, "meshes", "cube_cross.msh"), degree=2)
    ref = O.FEMMesh(m.elements(), m.vertices(), 2)
    assert (m.numVertices(), m.numElements(), m.numNodes()) == (64, 132, ref.num_nodes)
    assert np.array_equal(m.elementNodes(), ref.elem_nodes) and np.allclose(m.nodes(), ref.node_pos)
    assert np.array_equal(m.boundaryNodes(), ref.bdry_nodes)
    assert m.degree == 2 and m.simplexDimension == 3 and m.embeddingDimension == 3
    assert abs(m.volume - ref.embeddings_batch()[0].sum()) < 1e-13 and abs(m.bbox_volume - 8.0) < 1e-13
    pc = cmesh.PeriodicCondition(m)
    assert np.array_equal(pc.periodicDoFsForNodes(), O.periodic_dofs_for_nodes(ref)[0])
    m2 = cmesh.Mesh(os.path.join(GOLD, "meshes", "2D_microstructure.msh"), degree=1, embeddingDimension=2)
    assert m2.embeddingDimension == 2 and m2.vertices().shape[1] == 2 and m2.elementVolumes().min() > 0
    V, F = m2.vertices(), m2.elements()
    assert cmesh.Mesh(V, F, 2, 2).numNodes() == O.FEMMesh(F, V, 2).num_nodes


@pytest.mark.gpu
def test_spsd_system_on_caller_matrix():
    """sparse_matrices.SPSDSystem(K).fixVariables / solve on an arbitrary SPD matrix (a 2D 9-point-like
    stencil + random SPD perturbation), against scipy's direct solve."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    rng = np.random.default_rng(0)
    n = 400
    B = sp.random(n, n, density=0.01, random_state=1, format="csr")
    A = (B @ B.T + sp.diags(np.full(n, 0.5)) + sp.diags([-0.1] * (n - 1), 1) + sp.diags([-0.1] * (n - 1), -1)).tocoo()
    K = csm.TripletMatrix(n, n)
    for i, j, v in zip(A.row, A.col, A.data):
        if i <= j:
            K.addNZ(int(i), int(j), float(v) * 0.5)      # split into two halves: SPSDSystem must sum repeats
            K.addNZ(int(i), int(j), float(v) * 0.5)
    K.symmetry_mode = "UPPER_TRIANGLE"
    sysm = csm.SPSDSystem(K)
    Afull = A.tocsr()
    assert np.abs(sysm.ctx.export_scipy() - Afull).max() < 1e-13 * np.abs(Afull).max()
    b = rng.standard_normal(n)
    x = sysm.solve(b)
    assert np.linalg.norm(x - spl.spsolve(Afull.tocsc(), b)) < 1e-8 * np.linalg.norm(x)
    # fixVariables: reduced system with the fixed values moved to the right-hand side (SparseMatrices.hh:2457-2470)
    fv = np.array([3, 77, 150, 399])
    fx = np.array([0.5, -1.0, 0.0, 2.0])
    sysm.fixVariables(fv.tolist(), fx.tolist())
    x = sysm.solve(b)
    free = np.setdiff1d(np.arange(n), fv)
    xr = np.zeros(n); xr[fv] = fx
    rhs = b[free] - Afull[free][:, fv] @ fx
    xr[free] = spl.spsolve(Afull[free][:, free].tocsc(), rhs)
    assert np.array_equal(x[fv], fx)
    assert np.linalg.norm(x - xr) < 1e-8 * np.linalg.norm(xr)
    with pytest.raises(Exception):
        sysm.fixVariables([3], [1.0])                    # "Variable already fixed."
    with pytest.raises(RuntimeError):
        csm.SPSDSystem(K, C=csm.TripletMatrix(2, n + 1))   # wrong number of columns
    # constraint rows C x = C_rhs (SPSDSystem(K, C, C_rhs), the reference's UMFPACK / KKT branch): k + 1 SPD solves;
    # checked against a dense solve of [[K, C^T], [C, 0]] with the fixed variables eliminated
    Cd = np.zeros((3, n))
    Cd[0, :] = 1.0                                         # mean value
    Cd[1, ::7] = rng.standard_normal(len(Cd[1, ::7]))
    Cd[2, 5], Cd[2, 77] = 1.0, -2.0                        # touches a fixed variable: its value moves to the right-hand side
    crhs = np.array([0.3, -1.0, 0.5])
    C = csm.TripletMatrix(3, n)
    for r, cidx in zip(*np.nonzero(Cd)):
        C.addNZ(int(r), int(cidx), float(Cd[r, cidx]))
    sysc = csm.SPSDSystem(K, C, crhs)
    sysc.fixVariables(fv.tolist(), fx.tolist())
    Ad = Afull.toarray()
    kkt = np.block([[Ad[np.ix_(free, free)], Cd[:, free].T], [Cd[:, free], np.zeros((3, 3))]])
    for rhs_vec in (b, rng.standard_normal(n)):
        xc = sysc.solve(rhs_vec)
        sol = np.linalg.solve(kkt, np.concatenate([rhs_vec[free] - Ad[np.ix_(free, fv)] @ fx, crhs - Cd[:, fv] @ fx]))
        ref = np.zeros(n); ref[fv] = fx; ref[free] = sol[:len(free)]
        assert np.array_equal(xc[fv], fx) and np.abs(Cd @ xc - crhs).max() < 1e-7
        assert np.linalg.norm(xc - ref) < 1e-7 * np.linalg.norm(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("name,dim,deg", [("cube_cross", 3, 2), ("2D_microstructure", 2, 2), ("2D_microstructure", 2, 1)])
def test_periodic_homogenization_module(name, dim, deg):
    g = np.load(os.path.join(GOLD, "example_meshes.npz"))
    m = cmesh.Mesh(os.path.join(GOLD, "meshes", name + ".msh"), degree=deg, embeddingDimension=dim)
    Cbase = (ctensors.ElasticityTensor3D if dim == 3 else ctensors.ElasticityTensor2D)(200.0, 0.35)
    hr = cph.homogenize(m, Cbase)
    key = "%s_hom_p%d_" % (name, deg)
    # displacement form == stress form for the exact discrete solution (both in the golden's oracle)
    assert np.abs(hr.Ch.D - g[key + "Ch"]).max() < 1e-7 * np.abs(g[key + "Ch"]).max()
    for k in range(len(hr.w_ij)):
        assert np.abs(hr.w_ij[k].mean(axis=0)).max() < 1e-12                  # centred
        wg = g[key + "w"][k]
        assert np.linalg.norm(hr.w_ij[k] - (wg - wg.mean(axis=0))) < 1e-6 * np.linalg.norm(wg)
    # probe: u = E x + w with the face-average translation removed; strain = E + strain(w)
    fl = 6 if dim == 3 else 3
    ms = np.zeros(fl); ms[0] = 0.01; ms[fl - 1] = 0.005
    u, su = cph.probe(m, hr, ctensors.SymmetricMatrix(ms))
    assert u.shape == (m.numNodes(), dim) and su.shape == (m.numElements(), fl)
    vol = m.elementVolumes()
    sim = O.Simulator(m.elements(), m.vertices(), deg)
    avg = (vol[:, None] * sim.averageStrainField(u)).sum(axis=0) / vol.sum()
    assert np.abs(avg - (vol[:, None] * su).sum(axis=0) / vol.sum()).max() < 1e-10
    u2, _ = cph.probe(m, Cbase, ms)
    assert np.abs(u2 - u).max() < 1e-7 * np.abs(u).max()


@pytest.mark.gpu
@pytest.mark.parametrize("dim,deg", [(3, 1), (3, 2), (2, 2)])
def test_differential_operators_module(dim, deg):
    """`differential_operators.{laplacian, mass, mass_elasticity, bilaplacian, gradient}` (differential_operators.cc:21-90)
    against the oracle's Laplacian.hh / MassMatrix.hh restatement, incl. forceP1 on a quadratic mesh (vertex operators)."""
    import differential_operators as cdo
    import scipy.sparse as sp
    if dim == 3:
        V, T = O.grid_tet_mesh(2, 2, 1)
    else:
        V, Q = O.gen_grid_2d(3, 2)
        V, T = O.quad_tri_subdiv(V, Q)
        V = V[:, :2]
    m = cmesh.Mesh(V, T, degree=deg, embeddingDimension=dim)

    def dense(tm):
        return sp.coo_matrix((tm.arrays()[2], (tm.arrays()[0].astype(np.int64), tm.arrays()[1].astype(np.int64))), shape=(tm.m, tm.n)).toarray()

    def oracle_dense(tr, full=True):
        A = sp.coo_matrix((tr.v, (tr.i, tr.j)), shape=(tr.m, tr.n)).toarray()
        return A + np.triu(A, 1).T if full else A
    for forceP1 in (False, True):
        om = O.FEMMesh(T, V, 1 if forceP1 else deg)
        Lr, Mr = oracle_dense(O.laplacian_triplets(om)), oracle_dense(O.mass_triplets(om))
        Lg = cdo.laplacian(m, forceP1=forceP1)
        assert Lg.m == om.num_nodes and np.abs(dense(Lg) - Lr).max() < 1e-12 * np.abs(Lr).max()
        Lu = cdo.laplacian(m, forceP1=forceP1, upperTriOnly=True)
        assert Lu.symmetry_mode == "UPPER_TRIANGLE" and np.abs(dense(Lu) - np.triu(Lr)).max() < 1e-12 * np.abs(Lr).max()
        assert np.abs(dense(cdo.mass(m, forceP1=forceP1)) - Mr).max() < 1e-13 * np.abs(Mr).max()
        Ml = cdo.mass(m, lumped=True, forceP1=forceP1)
        assert np.abs(dense(Ml) - np.diag(Mr.sum(axis=1))).max() < 1e-13 * np.abs(Mr).max()
        Me = dense(cdo.mass_elasticity(m, forceP1=forceP1))
        assert Me.shape == (dim * om.num_nodes,) * 2
        for c in range(dim):
            assert np.abs(Me[c::dim, c::dim] - Mr).max() < 1e-13 * np.abs(Mr).max()
        assert np.abs(Me[0::dim, 1::dim]).max() == 0
        if np.abs(Mr.sum(axis=1)).min() > 1e-12:          # P2 triangles: int phi_vertex = 0, the lumped mass is singular
            B = cdo.bilaplacian(m, forceP1=forceP1).toarray()
            Br = Lr @ np.diag(1.0 / Mr.sum(axis=1)) @ Lr
            assert np.abs(B - Br).max() < 1e-11 * np.abs(Br).max()
    if deg == 1:
        g = cdo.gradient(m, V @ np.arange(1.0, dim + 1))               # linear field: constant gradient
        assert np.abs(g - np.arange(1.0, dim + 1)).max() < 1e-12
    else:
        with pytest.raises(RuntimeError, match="unimplemented"):
            cdo.gradient(m, np.zeros(m.numNodes()))


def test_benchmark_module_timers():
    import benchmark as cb
    cb.reset()
    cb.start_timer_section("Simulation"); cb.start_timer("Assemble System"); cb.stop_timer("Assemble System")
    cb.stop_timer_section("Simulation")
    d = cb.to_dict()
    assert set(d) == {"Simulation"} and "Assemble System" in d["Simulation"][1] and d["Simulation"][0] >= d["Simulation"][1]["Assemble System"] >= 0
    with pytest.raises(RuntimeError):
        cb.stop_timer_section("Simulation")
    import io
    buf = io.StringIO(); cb.report(out=buf)
    assert "Simulation" in buf.getvalue() and "Assemble System" in buf.getvalue()


def _spd_triplets(n, seed=2):
    import scipy.sparse as sp
    B = sp.random(n, n, density=0.02, random_state=seed, format="csr")
    A = (B @ B.T + sp.diags(np.full(n, 1.0))).tocoo()
    K = csm.TripletMatrix(n, n)
    for i, j, v in zip(A.row, A.col, A.data):
        if i <= j:
            K.addNZ(int(i), int(j), float(v))
    K.symmetry_mode = "UPPER_TRIANGLE"
    return K, A.tocsr()


def test_suite_sparse_matrix_container(tmp_path):
    """SuiteSparseMatrix (sparse_matrices.cc:67-141): CSC from triplets, apply with symmetric storage, trace, binary dump
    round trip (SparseMatrices.hh:1448-1495), pickling."""
    import pickle
    K, A = _spd_triplets(60)
    S = csm.SuiteSparseMatrix(K)
    assert (S.m, S.n) == (60, 60) and S.nz == len(S.Ax) == S.Ap[-1] and S.symmetry_mode == "UPPER_TRIANGLE"
    x = np.random.default_rng(0).standard_normal(60)
    assert np.abs(S.apply(x) - A @ x).max() < 1e-12 and abs(S.trace() - A.diagonal().sum()) < 1e-12
    p = str(tmp_path / "S.bin")
    S.dumpBinary(p)
    R = csm.SuiteSparseMatrix(p)
    for a, b in ((R.Ap, S.Ap), (R.Ai, S.Ai), (R.Ax, S.Ax)):
        assert np.array_equal(a, b)
    assert R.symmetry_mode == "UPPER_TRIANGLE"
    raw = np.fromfile(p, dtype=np.int64, count=3)
    assert raw.tolist() == [60, 60, S.nz] and os.path.getsize(p) == 3 * 8 + 4 + 61 * 8 + S.nz * 16
    P = pickle.loads(pickle.dumps(S))
    assert np.array_equal(P.Ax, S.Ax) and P.symmetry_mode == S.symmetry_mode
    T = S.getTripletMatrix()
    assert T.nnz == S.nz and np.abs(csm.SuiteSparseMatrix(T).toSciPy() - S.toSciPy()).max() == 0
    full = csm.SuiteSparseMatrix(K); full.symmetry_mode = "NONE"
    with pytest.raises(RuntimeError, match="Only symmetric"):
        full.solve(x)


@pytest.mark.gpu
def test_suite_sparse_matrix_solve():
    import scipy.sparse.linalg as spl
    K, A = _spd_triplets(300, seed=5)
    b = np.random.default_rng(1).standard_normal(300)
    x = csm.SuiteSparseMatrix(K).solve(b)
    assert np.linalg.norm(x - spl.spsolve(A.tocsc(), b)) < 1e-8 * np.linalg.norm(x)


def test_msh_field_writer_and_parser_bindings(tmp_path):
    """mesh.MSHFieldWriter / mesh.MSHFieldParser (MSHFieldWriter_bindings.cc, MSHFieldParser_bindings.cc)."""
    V, T = O.grid_tet_mesh(2, 1, 1)
    rng = np.random.default_rng(0)
    p = str(tmp_path / "f.msh")
    w = cmesh.MSHFieldWriter(p, V, T)
    u, E, eps = rng.random((len(V), 3)), rng.random(len(T)), rng.random((len(T), 6))
    w.addField("u", u); w.addField("E", E[:, None], cmesh.MSHFieldWriter.DomainType.PER_ELEMENT); w.close()
    with pytest.raises(RuntimeError, match="Invalid field domain size"):
        cmesh.MSHFieldWriter(str(tmp_path / "g.msh"), V, T).addField("bad", np.zeros(len(V) + len(T) + 1))
    from meshfem_amd import mesh_io
    w2 = mesh_io.MSHFieldWriter(str(tmp_path / "h.msh"), V, T); w2.addField("strain", eps, "element"); w2.addElementNodeField("s", rng.random((len(T), 4, 6))); w2.close()
    fp = cmesh.MSHFieldParser(p)
    assert fp.meshDimension() == 3 and fp.meshDegree() == 1 and fp.numElements() == len(T) and fp.numVertices() == len(V)
    assert np.array_equal(fp.vertices(), V) and np.array_equal(fp.elements(), T)
    assert np.array_equal(fp.vectorField("u"), u) and np.array_equal(fp.scalarField("E", "PER_ELEMENT"), E)
    assert fp.vectorFieldNames() == ["u"] and fp.scalarFieldNames("PER_ELEMENT") == ["E"] and fp.scalarFieldNames("PER_NODE") == []
    with pytest.raises(RuntimeError):
        fp.scalarField("u")
    fh = cmesh.MSHFieldParser(str(tmp_path / "h.msh"))
    assert np.array_equal(fh.symmetricMatrixField("strain"), eps) and fh.symmetricMatrixInterpolantFieldNames() == ["s"]
