"""Randomised self-consistency check on unstructured meshes (Delaunay of random points, slivers filtered by volume only): the assembled K
(mfh_export_bsr) against the oracle on the small isotropic ones, the matrix-free operator against the assembled K, and the three preconditioners
against each other on a random load. Used by tests/test_gpu_fuzz_unstructured.py (a few seeds) and scripts/fuzz_unstructured.py (many)."""
import numpy as np


def random_mesh(seed):
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(seed)
    dim = 3 if rng.random() < 0.7 else 2
    deg = int(rng.integers(1, 3))
    npts = int(rng.integers(40, 4000 if dim == 3 else 3000))
    P = rng.random((npts, dim)) * (rng.random(dim) * 3 + 0.5)
    E = Delaunay(P).simplices.astype(np.int32)
    vol = np.linalg.det(P[E[:, 1:]] - P[E[:, :1]])
    flip = vol < 0
    E[flip, 0], E[flip, 1] = E[flip, 1].copy(), E[flip, 0].copy()
    vol = np.abs(vol)
    E = np.ascontiguousarray(E[vol > 1e-6 * vol.mean()])
    used = np.unique(E)
    remap = -np.ones(npts, np.int64)
    remap[used] = np.arange(len(used))
    return rng, dim, deg, remap[E].astype(np.int32), np.ascontiguousarray(P[used]), int(rng.integers(0, 3))


def check(seed):
    """Returns (ok, one-line report)."""
    import meshfem_amd as M
    from meshfem_amd import grid
    from oracle import meshfem_oracle as O
    rng, dim, deg, E, V, mat = random_mesh(seed)
    tag = "seed %d dim %d deg %d verts %d elems %d mat %d" % (seed, dim, deg, len(V), len(E), mat)
    c = M.Context(0)
    try:
        c.mesh_build(E, V, deg)
        if mat == 0:
            c.material_isotropic(200.0, 0.3)
        elif mat == 1:
            c.material_ortho_field(grid.synthetic_orthotropic_field(len(E), dim, seed))
        else:
            c.material_iso_field(50 + 300 * rng.random(len(E)), 0.1 + 0.3 * rng.random(len(E)))
        ext = V.max(axis=0)
        c.bc_dirichlet_box([-1e-9] + [-1e9] * (dim - 1), [0.12 * ext[0]] + [1e9] * (dim - 1), [0.0] * dim)
        c.assemble()
        K = c.export_scipy()
        x = rng.standard_normal(dim * c.n_dof)
        y = c.apply_K(x)
        e_apply = np.linalg.norm(y - K @ x) / np.linalg.norm(y)
        e_sym = abs(K - K.T).max() / abs(K).max()
        e_or = -1.0
        if len(E) < 3000 and mat == 0:
            sim = O.Simulator(E, V, deg)
            sim.set_material_constant(O.ElasticityTensor.isotropic(dim, 200.0, 0.3))
            Ko = sim.assembleStiffnessMatrix().sum_repeated().to_scipy_full_from_upper()
            e_or = abs(K - Ko).max() / abs(Ko).max()
        f = rng.standard_normal(dim * c.n_dof)
        us, its = {}, {}
        for name, pre in (("bj", M.PRECOND_BLOCK_JACOBI), ("tl", M.PRECOND_TWO_LEVEL), ("mg", M.PRECOND_MULTIGRID)):
            c.set_preconditioner(pre)
            us[name] = c.sim_solve(f=f, rtol=1e-10, maxit=200000).ravel()
            its[name] = c.last_info["iterations"]
        d1 = np.linalg.norm(us["tl"] - us["bj"]) / np.linalg.norm(us["bj"])
        d2 = np.linalg.norm(us["mg"] - us["bj"]) / np.linalg.norm(us["bj"])
        ok = e_apply < 1e-12 and e_sym < 1e-13 and e_or < 1e-12 and d1 < 1e-6 and d2 < 1e-6
        return ok, "%s apply %.1e sym %.1e oracle %.1e  tl %.1e mg %.1e  its %s %s" % (tag, e_apply, e_sym, e_or, d1, d2, its, c.precond_info()["note"][:50])
    finally:
        c.close()
