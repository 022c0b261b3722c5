"""Synthetic grid meshes: vectorised restatement of the reference's `tools/grid AxBxC -t`
(src/bin/tools/grid.cc:115-137 = gen_grid, filters/gen_grid.hh:51-92, followed by hex_tet_subdiv,
filters/hex_tet_subdiv.hh:24-107: 24 tets per hex around the cell centre and shared face centres).
Vertex and element ORDER are identical to the reference's loops (checked against the literal loop
restatement in oracle/ by tests/test_host_logic.py)."""
import numpy as np

HEX_FACES = np.array([(0, 3, 2, 1), (0, 4, 7, 3), (4, 5, 6, 7), (1, 2, 6, 5), (0, 1, 5, 4), (2, 3, 7, 6)])


def gen_grid_3d(sx, sy, sz, z0=0):
    """Corner vertices (c, r, s) and GMSH-ordered hexes; z0 offsets the slice coordinate (slabs)."""
    nC, nR, nS = sx, sy, sz
    s, r, c = np.meshgrid(np.arange(nS + 1), np.arange(nR + 1), np.arange(nC + 1), indexing="ij")
    verts = np.stack([c.ravel(), r.ravel(), s.ravel() + z0], axis=1).astype(np.float64)

    def idx(s_, r_, c_):
        return (nC + 1) * ((nR + 1) * s_ + r_) + c_
    s, r, c = np.meshgrid(np.arange(nS), np.arange(nR), np.arange(nC), indexing="ij")
    s, r, c = s.ravel(), r.ravel(), c.ravel()
    hexes = np.stack([idx(s, r, c), idx(s, r, c + 1), idx(s, r + 1, c + 1), idx(s, r + 1, c),
                      idx(s + 1, r, c), idx(s + 1, r, c + 1), idx(s + 1, r + 1, c + 1), idx(s + 1, r + 1, c)], axis=1)
    return verts, hexes.astype(np.int64)


def hex_tet_subdiv(verts, hexes):
    """24 tets per hex. New vertices are appended in creation order: for every hex its centre, then
    each not-yet-seen face centre in face order (hex_tet_subdiv.hh:71-103)."""
    nV, nH = len(verts), len(hexes)
    fv = hexes[:, HEX_FACES]                                   # [nH, 6, 4]
    key = fv.min(axis=2) * np.int64(nV) + fv.max(axis=2)       # a grid face is identified by its diagonal
    flat = key.ravel()
    _, first, inv = np.unique(flat, return_index=True, return_inverse=True)
    is_new = first[inv] == np.arange(flat.size)
    new_before = np.cumsum(is_new) - is_new                    # new faces strictly before this flat index
    hex_of = np.arange(flat.size) // 6
    created = nV + (hex_of + 1) + new_before                   # vertex id if this (hex, face) creates it
    fc = created[first[inv]].reshape(nH, 6)
    new_per_hex_before = new_before.reshape(nH, 6)[:, 0]
    hc = nV + np.arange(nH) + new_per_hex_before
    n_out = nV + nH + int(is_new.sum())
    out_v = np.empty((n_out, 3))
    out_v[:nV] = verts
    out_v[hc] = verts[hexes].sum(axis=1) / 8
    fcenters = 0.25 * verts[fv].sum(axis=2).reshape(-1, 3)
    out_v[created[is_new]] = fcenters[is_new]
    a = fv[:, :, [1, 2, 3, 0]]                                 # e[f[(v+1)%4]]
    b = fv                                                     # e[f[v]]
    tets = np.stack([a, b, np.broadcast_to(fc[:, :, None], a.shape),
                     np.broadcast_to(hc[:, None, None], a.shape)], axis=3).reshape(-1, 4)
    return out_v, tets


def grid_tet_mesh(sx, sy, sz, min_corner=None, max_corner=None):
    """tools/grid AxBxC -t [--minCorner --maxCorner]: returns (vertices [nV,3] f64, tets [nE,4] i64)."""
    v, h = gen_grid_3d(sx, sy, sz)
    if min_corner is not None:
        mn, mx = np.asarray(min_corner, float), np.asarray(max_corner, float)
        v = v * ((mx - mn) / np.array([sx, sy, sz], float)) + mn
    return hex_tet_subdiv(v, h)


def gen_grid_2d(sx, sy):
    r, c = np.meshgrid(np.arange(sy + 1), np.arange(sx + 1), indexing="ij")
    verts = np.stack([c.ravel(), r.ravel()], axis=1).astype(np.float64)

    def idx(r_, c_):
        return (sx + 1) * r_ + c_
    r, c = np.meshgrid(np.arange(sy), np.arange(sx), indexing="ij")
    r, c = r.ravel(), c.ravel()
    quads = np.stack([idx(r, c), idx(r, c + 1), idx(r + 1, c + 1), idx(r + 1, c)], axis=1)
    return verts, quads.astype(np.int64)


def grid_tri_mesh(sx, sy, min_corner=None, max_corner=None):
    """2D grid split into 4 CCW triangles per quad around the quad centre."""
    v, q = gen_grid_2d(sx, sy)
    if min_corner is not None:
        mn, mx = np.asarray(min_corner, float), np.asarray(max_corner, float)
        v = v * ((mx - mn) / np.array([sx, sy], float)) + mn
    nV, nQ = len(v), len(q)
    centers = v[q].sum(axis=1) / 4
    ci = nV + np.arange(nQ)
    tris = np.stack([q, q[:, [1, 2, 3, 0]], np.broadcast_to(ci[:, None], q.shape)], axis=2).reshape(-1, 3)
    return np.concatenate([v, centers]), tris


def _morton(q):
    """Interleave the bits of the integer columns of q (<= 21 bits each) into one uint64 key."""
    def spread(x):
        x = x.astype(np.uint64) & np.uint64(0x1FFFFF)
        x = (x | (x << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        x = (x | (x << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        x = (x | (x << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        x = (x | (x << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        x = (x | (x << np.uint64(2))) & np.uint64(0x1249249249249249)
        return x
    key = np.zeros(len(q), dtype=np.uint64)
    for a in range(q.shape[1]):
        key |= spread(q[:, a]) << np.uint64(a)
    return key


def synthetic_orthotropic_field(n_elem, dim=3, seed=0):
    """BASELINE configs[3]'s per-element orthotropic field: E_* in U[100,300], nu_* in U[0.2,0.35], mu_* in U[40,120]
    (numpy default_rng(seed)), with the Poisson ratios of the rare draws whose compliance matrix is not positive definite
    (1 element in 2 million) scaled by 0.8 until it is -- SURVEY.md 8d: "check compliance PD". 3D: columns
    Ex,Ey,Ez,nuYX,nuZX,nuZY,muYZ,muZX,muXY; 2D: Ex,Ey,nuYX,muXY (ElasticityTensor.hh:136-164)."""
    rng = np.random.default_rng(seed)
    if dim == 3:
        P = np.column_stack([rng.uniform(100, 300, (n_elem, 3)), rng.uniform(0.2, 0.35, (n_elem, 3)), rng.uniform(40, 120, (n_elem, 3))])
        for _ in range(20):
            a00, a11, a22 = 1 / P[:, 0], 1 / P[:, 1], 1 / P[:, 2]
            a01, a02, a12 = -P[:, 3] / P[:, 1], -P[:, 4] / P[:, 2], -P[:, 5] / P[:, 2]
            m2 = a00 * a11 - a01 * a01                             # Sylvester: leading minors of the compliance block
            det = a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02)
            bad = ~((m2 > 1e-6 * a00 * a11) & (det > 1e-6 * a00 * a11 * a22))
            if not bad.any():
                break
            P[bad, 3:6] *= 0.8
        return P
    P = np.column_stack([rng.uniform(100, 300, (n_elem, 2)), rng.uniform(0.2, 0.35, n_elem), rng.uniform(40, 120, n_elem)])
    bad = P[:, 2] ** 2 >= P[:, 1] / P[:, 0]                     # 1/(Ex Ey) - (nuYX/Ey)^2 > 0
    P[bad, 2] = 0.5 * np.sqrt(P[bad, 1] / P[bad, 0])
    return P


def reorder_mesh(V, T, mode, seed=0):
    """Relabel vertices and permute elements: 'shuffle' (uniformly random, the worst case for gather
    locality) or 'morton' (Z-order space-filling curve on vertex positions / element barycentres).
    The mesh is the same up to numbering; SURVEY.md section 8(d) asks for these variants next to the
    generator order."""
    V, T = np.asarray(V, dtype=np.float64), np.asarray(T)
    if mode == "shuffle":
        rng = np.random.default_rng(seed)
        pv, pe = rng.permutation(len(V)), rng.permutation(len(T))
    elif mode == "morton":
        lo, ext = V.min(axis=0), np.maximum(V.max(axis=0) - V.min(axis=0), 1e-300)
        quant = lambda P: np.minimum(((P - lo) / ext * 2097151.0).astype(np.int64), 2097151)
        pv = np.argsort(_morton(quant(V)), kind="stable")
        pe = np.argsort(_morton(quant(V[T].mean(axis=1))), kind="stable")
    else:
        raise ValueError("mode must be 'shuffle' or 'morton'")
    inv = np.empty(len(V), dtype=np.int64)
    inv[pv] = np.arange(len(V))
    return V[pv], inv[T[pe]].astype(T.dtype)
