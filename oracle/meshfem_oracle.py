"""
oracle/meshfem_oracle.py -- CPU restatement (numpy/scipy) of MeshFEM's per-element
stiffness assembly + sparse solve hot path.

*** THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE. ***
Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it.
The product (meshfem_amd/, libmeshfem_hip.so) never imports, links or calls anything here.

PINNING STATUS: the reference cannot be built or imported in the authoring container
(every hot-path header pulls <Eigen/Dense>; SparseMatrices.hh includes <cholmod.h>/<umfpack.h>
unconditionally; none are installed and there is no network).  The pieces of this oracle that
the reference's own tests pin are checked against those golden values in tests/test_oracle_*.py:
  * quadrature rules         <- tests/test_quadrature.cc:26-40,50-170   (exact monomial integrals)
  * shape functions / grads  <- tests/test_shape_functions.cc:14-66
  * Voigt flattening         <- tests/test_tensors.cc:4-27
  * triplet/CSC/matvec       <- tests/test_sparse_matrices.cc:7-160
  * interpolants             <- tests/test_interpolant.cc:28-66
  * mass matrix L2 identity  <- tests/test_mass.cc:6-45 (on the reference's ball.msh / square_hole.off)
  * material JSON samples    <- tests/test_materials.cc:28-90
The end-to-end result (perElementStiffness -> K -> CHOLMOD solve) is pinned by NO reference test
and CHOLMOD itself (SuiteSparse, conda pin 5.4.0) is absent: for that boundary this oracle is
"PARITY UNPINNED" -- it is anchored instead on exact mathematics (sympy rational Ke, patch tests,
rigid-mode null space, energy identity) and scipy.sparse.linalg.splu stands in for CHOLMOD (a general sparse LU of
the KKT matrix for UMFPACK in the constraint-row branch). Also restated here: Laplacian.hh / MassMatrix.hh / Poisson.hh,
both homogenized-tensor forms of PeriodicHomogenization.hh, OrthotropicHomogenization.hh (checked on the reference's
2D_microstructure_orthocell.msh / 2D_microstructure.msh pair), assembleConstrainedSystem's rigid-motion rows, and the discrete
shape derivatives (deltaPerElementStiffness & co. and homogenizedElasticityTensorDiscreteDifferential, pinned on finite
differences of this oracle's own operators), both node-matching algorithms of PeriodicBoundaryMatcher.hh (match,
matchPermittingMismatch; ignoreDims), strainField / stressField.

All `file:line` citations are relative to /root/reference/src/lib/MeshFEM/ unless noted.
"""
import numpy as np

# --------------------------------------------------------------------------------------
# Simplex tables                                                        Simplex.hh:15-47
# --------------------------------------------------------------------------------------
EDGE_START = (0, 1, 2, 0, 2, 1)     # Simplex.hh:43  edgeStartNode
EDGE_END = (1, 2, 0, 3, 3, 3)       # Simplex.hh:44  edgeEndNode


def num_vertices(K):
    return K + 1


def num_edges(K):
    return (K * (K + 1)) // 2


def num_nodes(K, deg):              # Simplex.hh:24-29
    if K == 1:
        return deg + 1
    if K == 2:
        return ((deg + 1) * (deg + 2)) // 2
    if K == 3:
        return ((deg + 1) * (deg + 2) * (deg + 3)) // 6
    raise ValueError("Simplex dimension must be 1, 2, or 3")


# --------------------------------------------------------------------------------------
# Symmetric index flattening                                         Flattening.hh:19-83
# --------------------------------------------------------------------------------------
def flat_len(dim):
    return (dim * (dim + 1)) // 2


def flatten_indices(dim, i, j):     # Flattening.hh:23-27 (generic) == :47-60 (optimized)
    if i == j:
        return i
    if i < j:
        return (dim * (dim + 1) - j * (j - 1)) // 2 - (i + 1)
    return (dim * (dim + 1) - i * (i - 1)) // 2 - (j + 1)


def unflatten_index(dim, k):        # Flattening.hh:62-83
    if dim == 1:
        return (0, 0)
    if dim == 2:
        return (k, k) if k < 2 else (0, 1)
    if dim == 3:
        if k < 3:
            return (k, k)
        return {3: (1, 2), 4: (0, 2), 5: (0, 1)}[k]
    raise ValueError(dim)


# --------------------------------------------------------------------------------------
# Elasticity tensor                                     ElasticityTensor.hh:100-164,274-277
# --------------------------------------------------------------------------------------
class ElasticityTensor:
    """Flattened rank-4 tensor D (flatLen x flatLen, symmetric)."""

    def __init__(self, dim, D=None):
        self.dim = dim
        n = flat_len(dim)
        self.D = np.zeros((n, n)) if D is None else np.array(D, dtype=np.float64).reshape(n, n)

    @staticmethod
    def isotropic(dim, E, nu):      # ElasticityTensor.hh:100-134
        lam = (nu * E) / ((1.0 + nu) * (1.0 - 2.0 * nu))
        mu = E / (2.0 + 2.0 * nu)
        if dim == 2:                # plane stress
            lam = (nu * E) / (1.0 - nu * nu)
        return ElasticityTensor.isotropic_lame(dim, lam, mu)

    @staticmethod
    def isotropic_lame(dim, lam, mu):
        t = ElasticityTensor(dim)
        d = t.D
        for i in range(dim):
            for j in range(dim):
                d[i, j] = lam
            d[i, i] = lam + 2 * mu
        for k in range(dim, flat_len(dim)):
            d[k, k] = mu
        return t

    @staticmethod
    def orthotropic3d(Ex, Ey, Ez, nuYX, nuZX, nuZY, muYZ, muZX, muXY):   # :136-152
        m = np.zeros((6, 6))
        m[0, 0], m[0, 1], m[0, 2] = 1.0 / Ex, -nuYX / Ey, -nuZX / Ez
        m[1, 1], m[1, 2] = 1.0 / Ey, -nuZY / Ez
        m[2, 2] = 1.0 / Ez
        m[3, 3], m[4, 4], m[5, 5] = 1.0 / muYZ, 1.0 / muZX, 1.0 / muXY
        m = np.triu(m) + np.triu(m, 1).T
        return ElasticityTensor(3, np.linalg.inv(m))

    @staticmethod
    def orthotropic2d(Ex, Ey, nuYX, muXY):                               # :154-164
        m = np.zeros((3, 3))
        m[0, 0], m[0, 1] = 1.0 / Ex, -nuYX / Ey
        m[1, 1] = 1.0 / Ey
        m[2, 2] = 1.0 / muXY
        m = np.triu(m) + np.triu(m, 1).T
        return ElasticityTensor(2, np.linalg.inv(m))

    def __call__(self, i, j, k, l):  # :274-277
        return self.D[flatten_indices(self.dim, i, j), flatten_indices(self.dim, k, l)]

    def rank4(self):
        n = self.dim
        C = np.empty((n, n, n, n))
        for i in range(n):
            for j in range(n):
                for k in range(n):
                    for l in range(n):
                        C[i, j, k, l] = self(i, j, k, l)
        return C

    def double_contract_flat(self, e_flat):   # :444-449  D * shearDoubled(e)
        e = np.array(e_flat, dtype=np.float64).copy()
        e[self.dim:] *= 2.0
        return self.D @ e

    def double_contract(self, eps):           # symmetric matrix in, symmetric matrix out
        return unflatten_sym(self.dim, self.double_contract_flat(flatten_sym(self.dim, eps)))


def flatten_sym(dim, M):
    return np.array([M[unflatten_index(dim, k)] for k in range(flat_len(dim))])


def unflatten_sym(dim, v):
    M = np.zeros((dim, dim))
    for k in range(flat_len(dim)):
        i, j = unflatten_index(dim, k)
        M[i, j] = M[j, i] = v[k]
    return M


def canonical_strain(dim, k):       # SymmetricMatrix.hh:405-413: e_ij has 1 on diag, 1/2 shear
    v = np.zeros(flat_len(dim))
    v[k] = 1.0 if k < dim else 0.5
    return unflatten_sym(dim, v)


# --------------------------------------------------------------------------------------
# Gauss quadrature                                 GaussQuadrature.hh:37-59,115-192,283-340
# points are barycentric; weights sum to 1 (multiply by simplex volume)
# --------------------------------------------------------------------------------------
def _perm_rows(vals):
    return np.array(vals, dtype=np.float64)


def quadrature_rule(K, deg):
    """Return (points [nq, K+1], weights [nq]) of the rule the reference uses for polynomial
    degree `deg` on a K-simplex. Point ORDER follows the evaluation order in the reference."""
    if K == 1:                                                       # :37-59
        if deg <= 1:
            return _perm_rows([[0.5, 0.5]]), np.array([1.0])
        if deg <= 3:
            c0, c1 = 0.78867513459481288225, 0.21132486540518711775
            return _perm_rows([[c0, c1], [c1, c0]]), np.array([0.5, 0.5])
        if deg <= 5:
            c0, c1 = 0.11270166537925831148, 0.88729833462074168852
            return (_perm_rows([[c0, c1], [c1, c0], [0.5, 0.5]]),
                    np.array([5.0 / 18.0, 5.0 / 18.0, 4.0 / 9.0]))
    if K == 2:                                                       # :115-192
        third = 1 / 3.0
        if deg <= 1:
            return _perm_rows([[third] * 3]), np.array([1.0])
        if deg == 2:
            c0, c1 = 2 / 3.0, 1 / 6.0
            return _perm_rows([[c0, c1, c1], [c1, c0, c1], [c1, c1, c0]]), np.full(3, 1 / 3.0)
        if deg == 3:
            c0, c1 = 3 / 5.0, 1 / 5.0
            return (_perm_rows([[c0, c1, c1], [c1, c0, c1], [c1, c1, c0], [third] * 3]),
                    np.array([25.0 / 48] * 3 + [-9.0 / 16]))
        if deg == 4:
            w0, a0, b0 = 0.22338158967801146570, 0.10810301816807022736, 0.44594849091596488632
            w1, a1, b1 = 0.10995174365532186764, 0.81684757298045851308, 0.09157621350977074346
            return (_perm_rows([[a0, b0, b0], [b0, a0, b0], [b0, b0, a0],
                                [a1, b1, b1], [b1, a1, b1], [b1, b1, a1]]),
                    np.array([w0] * 3 + [w1] * 3))
        if deg == 5:
            w0, a0, b0 = 0.12593918054482715260, 0.79742698535308732240, 0.10128650732345633880
            w1, a1, b1 = 0.13239415278850618074, 0.059715871789769820459, 0.47014206410511508977
            return (_perm_rows([[a0, b0, b0], [b0, a0, b0], [b0, b0, a0],
                                [a1, b1, b1], [b1, a1, b1], [b1, b1, a1], [third] * 3]),
                    np.array([w0] * 3 + [w1] * 3 + [9.0 / 40]))
    if K == 3:                                                       # :283-340
        q = 0.25
        if deg <= 1:
            return _perm_rows([[q] * 4]), np.array([1.0])
        if deg == 2:
            c0, c1 = 0.58541019662496845446, 0.13819660112501051518
            return (_perm_rows([[c0, c1, c1, c1], [c1, c0, c1, c1], [c1, c1, c0, c1], [c1, c1, c1, c0]]),
                    np.full(4, 0.25))
        if deg == 3:
            c0, c1 = 0.5, 1 / 6.0
            return (_perm_rows([[c0, c1, c1, c1], [c1, c0, c1, c1], [c1, c1, c0, c1], [c1, c1, c1, c0],
                                [q] * 4]),
                    np.array([0.45] * 4 + [-0.8]))
        if deg == 4:
            a0, b0 = 11.0 / 14.0, 1.0 / 14.0
            a1, b1 = 0.39940357616679920500, 0.10059642383320079500
            pts = [[q] * 4,
                   [a0, b0, b0, b0], [b0, a0, b0, b0], [b0, b0, a0, b0], [b0, b0, b0, a0],
                   [a1, a1, b1, b1], [a1, b1, a1, b1], [a1, b1, b1, a1],
                   [b1, a1, a1, b1], [b1, a1, b1, a1], [b1, b1, a1, a1]]
            w = [-148.0 / 1875.0] + [343.0 / 7500.0] * 4 + [56.0 / 375.0] * 6
            return _perm_rows(pts), np.array(w)
    raise ValueError("no rule for K=%d deg=%d" % (K, deg))


def integrate(K, deg, f, vol=1.0):
    """Quadrature<K,deg>::integrate(f, vol)  GaussQuadrature.hh:412-417; f takes barycentric pt."""
    pts, w = quadrature_rule(K, deg)
    acc = None
    for p, wi in zip(pts, w):
        v = wi * np.asarray(f(p), dtype=np.float64)
        acc = v if acc is None else acc + v
    return acc * vol


# --------------------------------------------------------------------------------------
# Shape functions                                                   Functions.hh:86-102
# --------------------------------------------------------------------------------------
def shape_functions(deg, K, x):
    x = np.asarray(x, dtype=np.float64)
    if deg == 1:
        return x.copy()
    if deg == 2:
        nv = K + 1
        out = np.empty(num_nodes(K, 2))
        out[:nv] = 2 * x * (x - 0.5)
        for e in range(num_edges(K)):
            out[nv + e] = 4 * x[EDGE_START[e]] * x[EDGE_END[e]]
        return out
    raise ValueError(deg)


def integrated_shape_functions(deg, K):
    """Exact integrals of the nodal shape functions over a unit-volume simplex
    (Functions.hh:246-318 nodal-integration formulas)."""
    n = num_nodes(K, deg)
    out = np.zeros(n)
    if deg == 1:
        out[:] = 1.0 / n
    elif deg == 2:
        if K == 1:
            out[:] = [1 / 6.0, 1 / 6.0, 4 / 6.0]
        elif K == 2:
            out[3:] = 1 / 3.0
        else:
            out[:4] = -1 / 20.0
            out[4:] = 4 / 20.0
    return out


def interpolant_integrate(K, deg, nodal, vol):
    """Interpolant<T,K,deg>::integrate  Functions.hh:238-318 (deg 0,1,2)."""
    nodal = np.asarray(nodal, dtype=np.float64)
    if deg == 0:
        return nodal[0] * vol
    w = integrated_shape_functions(deg, K)
    return np.tensordot(w, nodal, axes=(0, 0)) * vol


# --------------------------------------------------------------------------------------
# Linear embedding of simplices                                EmbeddedElement.hh:43-241
# --------------------------------------------------------------------------------------
def embed_tet(P):
    """P: 4x3. returns (vol, gradLambda 3x4 (col k = grad lambda_k)).  :211-231"""
    p0, p1, p2, p3 = P
    n0 = np.cross(p3 - p1, p2 - p1)
    vol6 = np.dot(p0 - p1, n0)
    g = np.empty((3, 4))
    g[:, 0] = n0 / vol6
    g[:, 1] = np.cross(p2 - p0, p3 - p0) / vol6
    g[:, 2] = np.cross(p3 - p0, p1 - p0) / vol6
    g[:, 3] = np.cross(p1 - p0, p2 - p0) / vol6
    return vol6 / 6.0, g


def embed_tri2d(P):
    """P: 3x2. returns (area, gradLambda 2x3).  :170-190"""
    p0, p1, p2 = P
    e0, e1, e2 = p2 - p1, p0 - p2, p1 - p0
    dA = e1[0] * e2[1] - e1[1] * e2[0]
    g = np.empty((2, 3))
    for k, e in enumerate((e0, e1, e2)):
        g[:, k] = np.array([-e[1], e[0]]) / dA
    return dA / 2.0, g


def embed_tri3d(P):
    """P: 3x3 (boundary face). returns (area, gradLambda 3x3, unit normal).  :128-149"""
    p0, p1, p2 = P
    e0, e1, e2 = p2 - p1, p0 - p2, p1 - p0
    n = np.cross(e1, e2)
    dA = np.linalg.norm(n)
    n = n / dA
    g = np.empty((3, 3))
    for k, e in enumerate((e0, e1, e2)):
        g[:, k] = np.cross(n, e) / dA
    return dA / 2.0, g, n


def embed_edge2d(P):
    """P: 2x2 (boundary edge in 2D). returns (length, normal = CCW-rotated edge).  :87-104"""
    e = P[1] - P[0]
    L = np.linalg.norm(e)
    return L, np.array([-e[1], e[0]]) / L


def embed(K, P):
    if K == 3:
        return embed_tet(P)
    if K == 2:
        return embed_tri2d(P)
    raise ValueError(K)


def grad_phi_nodal(deg, K, gl, i):
    """EmbeddedElement::gradPhi(i): nodal values of the degree-(deg-1) interpolant of grad phi_i.
    Returns array [numNodes(K,deg-1), N].   EmbeddedElement.hh:288-313"""
    nv = K + 1
    if deg == 1:
        return gl[:, i][None, :].copy()
    out = np.zeros((nv, gl.shape[0]))
    if i < nv:
        for j in range(nv):
            out[j] = -gl[:, i]
        out[i] *= -3
    else:
        e = i - nv
        out[EDGE_START[e]] = 4 * gl[:, EDGE_END[e]]
        out[EDGE_END[e]] = 4 * gl[:, EDGE_START[e]]
    return out


def grad_phis_at(deg, K, gl, x):
    """EmbeddedElement::gradPhis(x) -> N x numNodes.   EmbeddedElement.hh:315-332"""
    nv = K + 1
    x = np.asarray(x, dtype=np.float64)
    if deg == 1:
        return gl.copy()
    out = np.zeros((gl.shape[0], num_nodes(K, 2)))
    out[:, :nv] = gl * (4.0 * x - 1.0)[None, :]
    for e in range(num_edges(K)):
        s, t = EDGE_START[e], EDGE_END[e]
        out[:, nv + e] = 4 * (x[t] * gl[:, s] + x[s] * gl[:, t])
    return out


def eval_interpolant(K, deg, nodal, x):
    """Interpolant<T,K,deg>::operator()(x) for deg 0/1/2 (Functions.hh:512-616)."""
    nodal = np.asarray(nodal)
    if deg == 0:
        return nodal[0]
    phi = shape_functions(deg, K, x)
    return np.tensordot(phi, nodal, axes=(0, 0))


# --------------------------------------------------------------------------------------
# Per-element stiffness                                       LinearElasticity.hh:165-232
# --------------------------------------------------------------------------------------
def per_element_stiffness_loop(deg, K, gl, vol, C):
    """Literal restatement of Element::perElementStiffness (upper triangle only; the lower
    triangle is returned as NaN exactly because the reference leaves it uninitialised)."""
    N = gl.shape[0]
    n = num_nodes(K, deg)
    Ke = np.full((n * N, n * N), np.nan)
    gp = [grad_phi_nodal(deg, K, gl, a) for a in range(n)]          # :195-197
    qdeg = 2 * (deg - 1)
    M = np.empty((N, N))
    for c in range(N):
        for d in range(c, N):
            for a in range(N):
                for b in range(N):
                    M[a, b] = C(a, c, d, b)                          # :203-205
            for j in range(n):
                vj = j * N + d
                Mgpj = gp[j] @ M.T                                   # :211-213  (M * grad at each node)
                for i in range(n):
                    vi = i * N + c
                    if c == d and vi > vj:
                        continue                                     # :219
                    val = integrate(K, qdeg, lambda p: np.dot(eval_interpolant(K, deg - 1, gp[i], p),
                                                              eval_interpolant(K, deg - 1, Mgpj, p)), vol)
                    if vi <= vj:
                        Ke[vi, vj] = val
                    else:
                        Ke[vj, vi] = val
    return Ke


def gradphi_at_quadrature(deg, K, gl_batch):
    """gl_batch [nE, N, K+1] -> G [nE, nq, n, N] (grad phi_i at the quadrature points of the
    degree-2(deg-1) rule) and the weights [nq]."""
    pts, w = quadrature_rule(K, 2 * (deg - 1))
    nE, N, nv = gl_batch.shape
    n = num_nodes(K, deg)
    G = np.zeros((nE, len(w), n, N))
    for q, p in enumerate(pts):
        if deg == 1:
            G[:, q, :, :] = np.transpose(gl_batch, (0, 2, 1))
        else:
            G[:, q, :nv, :] = np.transpose(gl_batch, (0, 2, 1)) * (4 * p - 1)[None, :, None]
            for e in range(num_edges(K)):
                s, t = EDGE_START[e], EDGE_END[e]
                G[:, q, nv + e, :] = 4 * (p[t] * gl_batch[:, :, s] + p[s] * gl_batch[:, :, t])
    return G, w


def per_element_stiffness_batch(deg, K, gl_batch, vol_batch, C4_batch):
    """Vectorised full symmetric Ke for many elements: formula A6 of SURVEY.md
    (== LinearElasticity.hh:183-231 summed out).  C4_batch: [nE or 1, N,N,N,N] rank-4 tensors.
    Returns [nE, n*N, n*N] with local dof index N*node+comp (LinearElasticity.hh:210,215)."""
    G, w = gradphi_at_quadrature(deg, K, gl_batch)
    nE, nq, n, N = G.shape
    # Ke[(i,c),(j,d)] = sum_q w_q vol sum_ab G[q,i,a] C[a,c,d,b] G[q,j,b]
    H = np.einsum('q,eqia,eqjb->eiajb', w, G, G, optimize=True)
    Ke = np.einsum('eiajb,eacdb->eicjd', H, np.broadcast_to(C4_batch, (nE, N, N, N, N)), optimize=True)
    Ke = Ke * vol_batch[:, None, None, None, None]
    return Ke.reshape(nE, n * N, n * N)


# --------------------------------------------------------------------------------------
# Synthetic meshes: tools/grid AxBxC -t      filters/gen_grid.hh:51-92, hex_tet_subdiv.hh:24-107
# --------------------------------------------------------------------------------------
HEX_FACES = ((0, 3, 2, 1), (0, 4, 7, 3), (4, 5, 6, 7), (1, 2, 6, 5), (0, 1, 5, 4), (2, 3, 7, 6))


def gen_grid_3d(sx, sy, sz):
    """gen_grid(sx,sy,sz): vertices (c,r,s) and GMSH-ordered hexes.  gen_grid.hh:51-92"""
    nC, nR, nS = sx, sy, sz
    s, r, c = np.meshgrid(np.arange(nS + 1), np.arange(nR + 1), np.arange(nC + 1), indexing='ij')
    verts = np.stack([c.ravel(), r.ravel(), s.ravel()], axis=1).astype(np.float64)

    def idx(s_, r_, c_):
        return (nC + 1) * ((nR + 1) * s_ + r_) + c_
    s, r, c = np.meshgrid(np.arange(nS), np.arange(nR), np.arange(nC), indexing='ij')
    s, r, c = s.ravel(), r.ravel(), c.ravel()
    hexes = np.stack([idx(s, r, c), idx(s, r, c + 1), idx(s, r + 1, c + 1), idx(s, r + 1, c),
                      idx(s + 1, r, c), idx(s + 1, r, c + 1), idx(s + 1, r + 1, c + 1), idx(s + 1, r + 1, c)],
                     axis=1)
    return verts, hexes


def hex_tet_subdiv(verts, hexes):
    """24 tets per hex; literal loop restatement of hex_tet_subdiv.hh:24-107 (small meshes)."""
    out_v = [v for v in verts]
    tets = []
    face_center = {}
    for e in hexes:
        hc = len(out_v)
        out_v.append(verts[e].sum(axis=0) / 8)
        for f in HEX_FACES:
            key = tuple(sorted(int(e[k]) for k in f))
            if key not in face_center:
                face_center[key] = len(out_v)
                out_v.append(0.25 * (verts[e[f[0]]] + verts[e[f[1]]] + verts[e[f[2]]] + verts[e[f[3]]]))
            fc = face_center[key]
            for v in range(4):
                tets.append((int(e[f[(v + 1) % 4]]), int(e[f[v]]), fc, hc))
    return np.array(out_v, dtype=np.float64), np.array(tets, dtype=np.int64)


def grid_tet_mesh(sx, sy, sz, min_corner=None, max_corner=None):
    """tools/grid AxBxC -t [--minCorner --maxCorner]  (src/bin/tools/grid.cc:115-137)."""
    v, h = gen_grid_3d(sx, sy, sz)
    if min_corner is not None:
        mn, mx = np.asarray(min_corner, float), np.asarray(max_corner, float)
        v = v * ((mx - mn) / np.array([sx, sy, sz], float)) + mn
    return hex_tet_subdiv(v, h)


def gen_grid_2d(sx, sy):
    """gen_grid(sx,sy): quads in GMSH order.  gen_grid.hh:20-49"""
    r, c = np.meshgrid(np.arange(sy + 1), np.arange(sx + 1), indexing='ij')
    verts = np.stack([c.ravel(), r.ravel()], axis=1).astype(np.float64)

    def idx(r_, c_):
        return (sx + 1) * r_ + c_
    r, c = np.meshgrid(np.arange(sy), np.arange(sx), indexing='ij')
    r, c = r.ravel(), c.ravel()
    quads = np.stack([idx(r, c), idx(r, c + 1), idx(r + 1, c + 1), idx(r + 1, c)], axis=1)
    return verts, quads


def quad_tri_subdiv(verts, quads):
    """4 triangles per quad around the quad centre (symmetric split used for 2D grids;
    filters/quad_tri_subdiv.hh). CCW triangles (corner k, corner k+1, centre)."""
    out_v = [v for v in verts]
    tris = []
    for q in quads:
        ci = len(out_v)
        out_v.append(verts[q].sum(axis=0) / 4)
        for k in range(4):
            tris.append((int(q[k]), int(q[(k + 1) % 4]), ci))
    return np.array(out_v, dtype=np.float64), np.array(tris, dtype=np.int64)


# --------------------------------------------------------------------------------------
# FEMMesh: node numbering + boundary       FEMMesh.inl:17-82, FEMMesh.hh:221-237,
#                                           TetMesh.inl:36-91, TriMesh.inl:86-118
# --------------------------------------------------------------------------------------
TET_FACE_CORNERS = ((1, 3, 2), (0, 2, 3), (0, 3, 1), (0, 1, 2))   # TetMesh.hh:221-226


class FEMMesh:
    def __init__(self, elems, verts, deg):
        elems = np.asarray(elems, dtype=np.int64)
        verts = np.asarray(verts, dtype=np.float64)
        self.K = elems.shape[1] - 1
        self.N = verts.shape[1]
        self.deg = deg
        self.elems = elems
        self.verts = verts
        nE, nV = len(elems), len(verts)
        self.num_vertices = nV
        K = self.K
        ne = num_edges(K)
        # --- edge nodes in first-encounter order over (element, local edge)  FEMMesh.inl:22-36
        if deg == 2:
            edge_of = {}
            elem_edge_node = np.empty((nE, ne), dtype=np.int64)
            for s in range(nE):
                for ei in range(ne):
                    a, b = int(elems[s, EDGE_START[ei]]), int(elems[s, EDGE_END[ei]])
                    key = (a, b) if a < b else (b, a)
                    k = edge_of.setdefault(key, len(edge_of))
                    elem_edge_node[s, ei] = k
            self.edge_of = edge_of
            self.num_edge_nodes = len(edge_of)
            self.elem_nodes = np.concatenate([elems, nV + elem_edge_node], axis=1)
            pos = np.empty((nV + len(edge_of), self.N))
            pos[:nV] = verts
            for (a, b), k in edge_of.items():
                pos[nV + k] = 0.5 * (verts[a] + verts[b])             # FEMMesh.hh:228-233
            self.node_pos = pos
        else:
            self.edge_of = {}
            self.num_edge_nodes = 0
            self.elem_nodes = elems.copy()
            self.node_pos = verts.copy()
        self.num_nodes = len(self.node_pos)
        self.nodes_per_elem = self.elem_nodes.shape[1]
        self._build_boundary()

    # boundary elements / vertices / nodes
    def _build_boundary(self):
        K, elems = self.K, self.elems
        owner = {}                                                     # boundary simplex -> its volume element
        if K == 3:
            faces = {}
            for t in range(len(elems)):
                for f in range(4):
                    vs = tuple(int(elems[t, c]) for c in TET_FACE_CORNERS[f])
                    key = tuple(sorted(vs))
                    if key in faces:
                        del faces[key]                                 # TetMesh.inl:59-68
                    else:
                        faces[key] = vs
                        owner[key] = t
            bverts_of_elem = []
            for key in sorted(faces.keys()):                           # std::map order :75
                vs = faces[key]
                # boundary face corner c = volume half-face corner 2-c  TetMesh.hh:463-469
                bverts_of_elem.append((vs[2], vs[1], vs[0]))
            # boundary vertex numbering: first encounter over VOLUME half-face corners 0..2 (:82-89)
            order_src = [faces[k] for k in sorted(faces.keys())]
        elif K == 2:
            edges = {}
            for t in range(len(elems)):
                for c in range(3):
                    tail, tip = int(elems[t, (c + 1) % 3]), int(elems[t, (c + 2) % 3])  # TriMesh.hh:285-298
                    key = (min(tail, tip), max(tail, tip))
                    if key in edges:
                        del edges[key]
                    else:
                        edges[key] = (tail, tip)
                        owner[key] = t
            bverts_of_elem = []
            order_src = []
            for key in sorted(edges.keys()):
                tail, tip = edges[key]
                # boundary edge tip = vol HE tail and vice versa (TriMesh.inl:98-100);
                # boundary edge vertex 0 = its tail, vertex 1 = its tip
                bverts_of_elem.append((tip, tail))
                order_src.append((tail, tip))                          # tipVV first, then tailVV (:103-104)
        else:
            raise ValueError(K)
        self.bdry_elem_verts = np.array(bverts_of_elem, dtype=np.int64).reshape(-1, K)
        self.bdry_parent = np.array([owner[k] for k in sorted(faces.keys() if K == 3 else edges.keys())], dtype=np.int64)
        bv_index = {}
        for vs in order_src:
            for v in vs:
                bv_index.setdefault(v, len(bv_index))
        self.bdry_vertices = np.array(list(bv_index.keys()), dtype=np.int64)   # volume vertex ids
        # boundary nodes: boundary vertices, then boundary edge nodes in first encounter order over
        # boundary simplices and their local edges (0,1),(1,2),(2,0)   FEMMesh.inl:43-58
        nbe = len(self.bdry_elem_verts)
        if self.deg == 2:
            nbedges = num_edges(K - 1)
            bedge_index = {}
            be_edge_nodes = np.empty((nbe, nbedges), dtype=np.int64)
            for b in range(nbe):
                vs = self.bdry_elem_verts[b]
                for ei in range(nbedges):
                    a, c = int(vs[EDGE_START[ei]]), int(vs[EDGE_END[ei]])
                    vol_edge = self.edge_of[(a, c) if a < c else (c, a)]
                    bedge_index.setdefault(vol_edge, len(bedge_index))
                    be_edge_nodes[b, ei] = self.num_vertices + vol_edge
            self.bdry_elem_nodes = np.concatenate([self.bdry_elem_verts, be_edge_nodes], axis=1)
            self.bdry_nodes = np.concatenate([self.bdry_vertices,
                                              self.num_vertices + np.array(list(bedge_index.keys()), dtype=np.int64)])
        else:
            self.bdry_elem_nodes = self.bdry_elem_verts.copy()
            self.bdry_nodes = self.bdry_vertices.copy()
        self.is_bdry_node = np.zeros(self.num_nodes, dtype=bool)
        self.is_bdry_node[self.bdry_nodes] = True

    # element geometry
    def embeddings(self):
        nE = len(self.elems)
        vol = np.empty(nE)
        gl = np.empty((nE, self.N, self.K + 1))
        for e in range(nE):
            vol[e], gl[e] = embed(self.K, self.verts[self.elems[e]])
        return vol, gl

    def embeddings_batch(self):
        """Vectorised EmbeddedElement.hh:211-231 / :170-190."""
        P = self.verts[self.elems]
        if self.K == 3:
            p0, p1, p2, p3 = P[:, 0], P[:, 1], P[:, 2], P[:, 3]
            n0 = np.cross(p3 - p1, p2 - p1)
            v6 = np.einsum('ij,ij->i', p0 - p1, n0)
            gl = np.stack([n0, np.cross(p2 - p0, p3 - p0), np.cross(p3 - p0, p1 - p0),
                           np.cross(p1 - p0, p2 - p0)], axis=2) / v6[:, None, None]
            return v6 / 6.0, gl
        p0, p1, p2 = P[:, 0], P[:, 1], P[:, 2]
        e = [p2 - p1, p0 - p2, p1 - p0]
        dA = e[1][:, 0] * e[2][:, 1] - e[1][:, 1] * e[2][:, 0]
        gl = np.stack([np.stack([-ek[:, 1], ek[:, 0]], axis=1) for ek in e], axis=2) / dA[:, None, None]
        return dA / 2.0, gl

    def bdry_elem_geometry(self):
        """volume (area/length) and outward normal of every boundary element."""
        nb = len(self.bdry_elem_verts)
        vol = np.empty(nb)
        nrm = np.empty((nb, self.N))
        for b in range(nb):
            P = self.verts[self.bdry_elem_verts[b]]
            if self.K == 3:
                vol[b], _, nrm[b] = embed_tri3d(P)
            else:
                vol[b], nrm[b] = embed_edge2d(P)
        return vol, nrm

    def bounding_box(self):
        return self.node_pos.min(axis=0), self.node_pos.max(axis=0)


# --------------------------------------------------------------------------------------
# Triplet matrices, sumRepeated, CSC            SparseMatrices.hh:191-773,818-1786
# --------------------------------------------------------------------------------------
class TripletMatrix:
    def __init__(self, m=0, n=0):
        self.m, self.n = m, n
        self.i = np.zeros(0, dtype=np.int64)
        self.j = np.zeros(0, dtype=np.int64)
        self.v = np.zeros(0, dtype=np.float64)

    @staticmethod
    def from_arrays(m, n, i, j, v):
        t = TripletMatrix(m, n)
        t.i, t.j, t.v = np.asarray(i, np.int64), np.asarray(j, np.int64), np.asarray(v, np.float64)
        return t

    def nnz(self):
        return len(self.v)

    def sum_repeated(self):
        """TripletMatrix::sumRepeated (SparseMatrices.hh:280-374): column-major sort, sum runs,
        drop exact zeros (pruneTol = 0, :211,370-373)."""
        if self.nnz() == 0:
            return self
        order = np.lexsort((self.i, self.j))
        i, j, v = self.i[order], self.j[order], self.v[order]
        new = np.ones(len(v), dtype=bool)
        new[1:] = (i[1:] != i[:-1]) | (j[1:] != j[:-1])
        starts = np.flatnonzero(new)
        vs = np.add.reduceat(v, starts)
        keep = vs != 0.0
        self.i, self.j, self.v = i[starts][keep], j[starts][keep], vs[keep]
        return self

    def to_csc(self):
        """getCompressedColumn (SparseMatrices.hh:422-447) after sumRepeated."""
        self.sum_repeated()
        Ap = np.zeros(self.n + 1, dtype=np.int64)
        np.add.at(Ap, self.j + 1, 1)
        Ap = np.cumsum(Ap)
        return Ap, self.i.copy(), self.v.copy()

    def dump_binary(self, path):
        """TripletMatrix::dumpBinary (SparseMatrices.hh:629-645): u64 nnz, u64 i[], u64 j[], f64 v[]."""
        with open(path, 'wb') as f:
            np.array([self.nnz()], dtype=np.uint64).tofile(f)
            self.i.astype(np.uint64).tofile(f)
            self.j.astype(np.uint64).tofile(f)
            self.v.astype(np.float64).tofile(f)

    def to_scipy_full_from_upper(self):
        import scipy.sparse as sp
        U = sp.coo_matrix((self.v, (self.i, self.j)), shape=(self.m, self.n)).tocsc()
        return U + sp.triu(U, 1).T

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.coo_matrix((self.v, (self.i, self.j)), shape=(self.m, self.n)).tocsc()


def csc_apply_symmetric_upper(Ap, Ai, Ax, x):
    """CSCMatrix::applyRaw for an UPPER-stored symmetric matrix (SparseMatrices.hh:1576-1592)."""
    y = np.zeros(len(Ap) - 1)
    for j in range(len(Ap) - 1):
        for k in range(Ap[j], Ap[j + 1]):
            i = Ai[k]
            y[i] += Ax[k] * x[j]
            if i != j:
                y[j] += Ax[k] * x[i]
    return y


# --------------------------------------------------------------------------------------
# SPSDSystem                                   SparseMatrices.hh:2321-2716
# --------------------------------------------------------------------------------------
class SPSDSystem:
    """Restatement of SPSDSystem for the SPD branch (no constraint rows). CHOLMOD is replaced by
    scipy.sparse.linalg.splu (stand-in, see module header)."""

    def __init__(self, K_upper: TripletMatrix):
        self.numVars = K_upper.m
        up = K_upper.i <= K_upper.j                                   # setUpperTriangle :2337
        self.A = TripletMatrix.from_arrays(K_upper.m, K_upper.n, K_upper.i[up], K_upper.j[up], K_upper.v[up])
        self.reducedVarForVar = np.arange(self.numVars, dtype=np.int64)
        self.fixedVarValues = np.zeros(0)
        self.fixedVarRHSContribution = np.zeros(self.numVars)
        self._lu = None

    def fix_variables(self, fixed_vars, fixed_values=None):           # :2389-2500
        fixed_vars = np.asarray(fixed_vars, dtype=np.int64)
        if len(fixed_vars) == 0:
            return
        fix_to_zero = fixed_values is None or len(fixed_values) == 0
        m = self.A.m
        replacement = np.zeros(m, dtype=np.int64)
        newly = np.zeros(m)
        if not fix_to_zero:
            fixed_values = np.asarray(fixed_values, dtype=np.float64)
            rv = self.reducedVarForVar[fixed_vars]
            ok = rv >= 0
            newly[rv[ok]] = fixed_values[ok]
        base = len(self.fixedVarValues)
        self.fixedVarValues = np.concatenate([self.fixedVarValues, np.zeros(len(fixed_vars))])
        for k, to_fix in enumerate(fixed_vars):
            curr = self.reducedVarForVar[to_fix]
            if curr < 0:
                raise RuntimeError("Variable already fixed.")
            replacement[curr] = -1
            self.reducedVarForVar[to_fix] = -1 - (base + k)
            if not fix_to_zero:
                self.fixedVarValues[base + k] = fixed_values[k]
        keep = replacement >= 0
        replacement[keep] = np.arange(keep.sum())
        cur = self.reducedVarForVar
        pos = cur >= 0
        cur[pos] = replacement[cur[pos]]
        if not fix_to_zero:                                           # :2457-2470
            ti, tj, tv = self.A.i, self.A.j, self.A.v
            np.subtract.at(self.fixedVarRHSContribution, ti, tv * newly[tj])
            strict = ti < tj
            np.subtract.at(self.fixedVarRHSContribution, tj[strict], tv[strict] * newly[ti[strict]])
        ri, rj = replacement[self.A.i], replacement[self.A.j]
        k2 = (ri >= 0) & (rj >= 0)
        self.A = TripletMatrix.from_arrays(m - len(fixed_vars), m - len(fixed_vars), ri[k2], rj[k2], self.A.v[k2])
        self.fixedVarRHSContribution = self.fixedVarRHSContribution[keep]
        self._lu = None

    def solve(self, f):                                               # :2515-2606
        import scipy.sparse.linalg as spla
        f = np.asarray(f, dtype=np.float64).ravel()
        if len(f) != self.numVars:
            raise RuntimeError("Bad RHS")
        b = np.zeros(self.A.m)
        free = self.reducedVarForVar >= 0
        b[self.reducedVarForVar[free]] = f[free] + self.fixedVarRHSContribution[self.reducedVarForVar[free]]
        if self._lu is None:
            self.A.sum_repeated()
            self._lu = spla.splu(self.A.to_scipy_full_from_upper().tocsc(), permc_spec='MMD_AT_PLUS_A',
                                 diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
        ur = self._lu.solve(b)
        u = np.empty(self.numVars)
        u[free] = ur[self.reducedVarForVar[free]]
        u[~free] = self.fixedVarValues[-1 - self.reducedVarForVar[~free]]
        return u


# --------------------------------------------------------------------------------------
# Periodic DoFs              BoundaryConditions.hh:452-561, PeriodicBoundaryMatcher.hh:111-260
# --------------------------------------------------------------------------------------
def periodic_dofs_for_nodes(mesh: FEMMesh, eps=1e-7, ignore_mismatch=False, ignore_dims=()):
    """PeriodicCondition (BoundaryConditions.hh:457-561): dofForNode + numDoFs + per-boundary-element isInternal flag.
    Nodes on opposite faces of the bounding-box cell are identified with PeriodicBoundaryMatcher::match (:149-260; a
    mismatch throws) or ::matchPermittingMismatch (:262-360); `ignore_dims` removes the memberships of the non-periodic
    dimensions (:470-500). DoF ids are assigned in volume-node order, every identified node receiving the id at the first
    one's turn (:533-554)."""
    pos = mesh.node_pos
    mn, mx = mesh.bounding_box()
    N = mesh.N
    periodic = np.array([d not in tuple(ignore_dims) for d in range(N)])
    on_min = (np.abs(pos - mn) <= eps) & periodic[None, :]
    on_max = (np.abs(pos - mx) <= eps) & periodic[None, :]
    bn = np.flatnonzero(mesh.is_bdry_node)
    P = pos[bn]
    bmin, bmax = on_min[bn], on_max[bn]

    def closest(query, cand):                                         # CollisionGrid::getClosestPoint(query, eps)
        if len(cand) == 0:
            return -1
        d = np.linalg.norm(P[cand] - query, axis=1)
        k = int(np.argmin(d))
        return int(cand[k]) if d[k] <= eps else -1

    nb = len(bn)
    set_for = np.full(nb, -1, dtype=np.int64)
    sets = []
    if not ignore_mismatch:                                           # match
        minimal = ~bmax.any(axis=1)
        nonmin = np.flatnonzero(~minimal)
        for i in range(nb):
            if not minimal[i]:
                continue
            set_for[i] = len(sets)
            dims = np.flatnonzero(bmin[i])
            ns = [i]
            for n in range(1, 1 << len(dims)):
                q = P[i].copy()
                for idx, d in enumerate(dims):
                    if n & (1 << idx):
                        q[d] = mx[d]
                r = closest(q, nonmin)
                if r < 0:
                    raise RuntimeError("Couldn't find %dth periodic-identified node for minimal boundary node %d at %s; looking for %s"
                                       % (n, i, P[i], q))
                if set_for[r] != -1:
                    raise RuntimeError("Non bijective node set assignment.")
                set_for[r] = set_for[i]
                ns.append(r)
            sets.append(ns)
        un = np.flatnonzero(set_for < 0)
        if len(un):
            raise RuntimeError("Unmatched non-minimal boundary node %d at %s" % (un[0], P[un[0]]))
    else:                                                             # matchPermittingMismatch
        pair = {}
        for d in range(N):
            cand = np.flatnonzero(bmin[:, d])
            for i in np.flatnonzero(bmax[:, d]):
                q = P[i].copy()
                q[d] = mn[d]
                pi = closest(q, cand)
                if pi < 0:
                    continue                                          # mismatch!
                a, b = pair.setdefault(int(i), [-1] * N), pair.setdefault(pi, [-1] * N)
                if a[d] != -1 or b[d] != -1:
                    raise RuntimeError("Non-bijective boundary matching")
                a[d], b[d] = pi, int(i)
        for i in range(nb):                                           # connected components of the pair graph
            if set_for[i] != -1:
                continue
            set_for[i] = len(sets)
            comp, queue = [i], [i]
            while queue:
                u = queue.pop(0)
                for v in pair.get(u, []):
                    if v != -1 and set_for[v] == -1:
                        set_for[v] = set_for[i]
                        comp.append(v)
                        queue.append(v)
            sets.append(comp)
    ident = {}
    for ns in sets:
        g = [int(bn[k]) for k in ns]
        for ni in g:
            ident[ni] = g
    dof = np.full(mesh.num_nodes, -1, dtype=np.int64)
    nd = 0
    for ni in range(mesh.num_nodes):
        if dof[ni] >= 0:
            continue
        if ni in ident:
            for nj in ident[ni]:
                dof[nj] = nd
        else:
            dof[ni] = nd
        nd += 1
    # boundary elements whose nodes all share one cell face are internal (PeriodicBoundaryMatcher.hh:127-145)
    internal = np.zeros(len(mesh.bdry_elem_nodes), dtype=bool)
    for b, nodes in enumerate(mesh.bdry_elem_nodes):
        fm_min = on_min[nodes].all(axis=0)
        fm_max = on_max[nodes].all(axis=0)
        internal[b] = bool(fm_min.any() or fm_max.any())
    return dof, nd, internal


# --------------------------------------------------------------------------------------
# LinearElasticity::Simulator                                   LinearElasticity.hh:434-1659
# --------------------------------------------------------------------------------------
class Simulator:
    def __init__(self, elems, verts, deg, mesh=None):
        self.mesh = mesh if mesh is not None else FEMMesh(elems, verts, deg)
        m = self.mesh
        self.N, self.K, self.deg = m.N, m.K, deg
        self.vol, self.gl = m.embeddings_batch()
        if (self.vol < 0).any():                                      # :465-472
            raise RuntimeError("Mesh has negatively oriented elements.")
        nE = len(m.elems)
        self.C4 = np.broadcast_to(ElasticityTensor.isotropic(self.N, 1.0, 0.3).rank4(), (1,) + (self.N,) * 4)
        self.D = [ElasticityTensor.isotropic(self.N, 1.0, 0.3)]      # Materials.hh:408 default
        self.dofForNode = None
        self.numDoFs_ = m.num_nodes
        nb = len(m.bdry_elem_nodes)
        self.neumannTraction = np.zeros((nb, self.N))
        self.beInternal = np.zeros(nb, dtype=bool)
        self.dirichletMask = np.zeros((m.num_nodes, self.N), dtype=bool)     # per (boundary) node
        self.dirichletValue = np.zeros((m.num_nodes, self.N))
        self.deltaForces = []
        self.useNRTPin = False
        self.useRigidMotionConstraint = False
        self._system = None

    # ---- materials
    def set_material_constant(self, tensor: ElasticityTensor):
        self.D = [tensor]
        self.C4 = tensor.rank4()[None]
        self._system = None

    def set_material_field(self, tensors):
        self.D = list(tensors)
        self.C4 = np.stack([t.rank4() for t in tensors])
        self._system = None

    def elem_D(self, e):
        return self.D[e if len(self.D) > 1 else 0]

    # ---- DoFs
    def numDoFs(self):
        return self.numDoFs_

    def DoF(self, node):                                              # :831-836
        return node if self.dofForNode is None else self.dofForNode[node]

    def dof_array(self):
        return np.arange(self.mesh.num_nodes) if self.dofForNode is None else self.dofForNode

    def applyPeriodicConditions(self, eps=1e-7, ignore_mismatch=False, ignore_dims=()):   # :845-854
        self.dofForNode, self.numDoFs_, self.beInternal = periodic_dofs_for_nodes(self.mesh, eps, ignore_mismatch, ignore_dims)
        self._system = None

    # ---- boundary conditions (box regions)                         :881-1027
    def apply_dirichlet_box(self, mn, mx, value, components=(True, True, True)):
        m = self.mesh
        mn, mx = np.asarray(mn, float), np.asarray(mx, float)
        for ni in m.bdry_nodes:
            p = m.node_pos[ni]
            if (p >= mn).all() and (p <= mx).all():                  # BBox::containsPoint inclusive
                for c in range(self.N):
                    if not components[c]:
                        continue
                    if not self.dirichletMask[ni, c]:
                        self.dirichletMask[ni, c] = True
                        self.dirichletValue[ni, c] = value[c]
                    elif abs(self.dirichletValue[ni, c] - value[c]) > 1e-10:
                        raise RuntimeError("Conflicting dirichlet displacements.")
        self._system = None

    def apply_neumann_box(self, mn, mx, value, kind='traction'):
        """kind: 'traction' | 'force' (total force / region area) | 'pressure' (scalar value)."""
        m = self.mesh
        mn, mx = np.asarray(mn, float), np.asarray(mx, float)
        bvol, bnrm = m.bdry_elem_geometry()
        region, area = [], 0.0
        for b, vs in enumerate(m.bdry_elem_verts):
            center = m.verts[vs].sum(axis=0) / len(vs)               # :903-906
            if (center >= mn).all() and (center <= mx).all():
                region.append(b)
                area += bvol[b]
                if kind == 'pressure':
                    self.neumannTraction[b] = -value * bnrm[b]
                else:
                    self.neumannTraction[b] = np.asarray(value, float)
        if not region:
            raise RuntimeError("Neumann region unmatched")
        if kind == 'force':
            for b in region:
                self.neumannTraction[b] = self.neumannTraction[b] / area
        self._system = None

    def box_percent(self, mn_rel, mx_rel):
        """'box%' regions are relative to the mesh bbox (BoundaryConditions.cc:310-316)."""
        bmn, bmx = self.mesh.bounding_box()
        d = bmx - bmn
        return bmn + np.asarray(mn_rel) * d, bmn + np.asarray(mx_rel) * d

    # ---- loads
    def neumannLoad(self):                                            # :703-717, :341-347
        m = self.mesh
        load = np.zeros((self.numDoFs(), self.N))
        bvol, _ = m.bdry_elem_geometry()
        w = integrated_shape_functions(self.deg, self.K - 1)
        for b, nodes in enumerate(m.bdry_elem_nodes):
            for n, node in enumerate(nodes):
                load[self.DoF(node)] += (w[n] * bvol[b]) * self.neumannTraction[b]
        for ni, f in self.deltaForces:
            load[self.DoF(ni)] += f
        return load

    def constantStrainLoad(self, cstrain):                            # :551-562, :135-162
        m = self.mesh
        load = np.zeros((self.numDoFs(), self.N))
        for e in range(len(m.elems)):
            cstress = self.elem_D(e).double_contract(cstrain)
            for i in range(m.nodes_per_elem):
                gint = interpolant_integrate(self.K, self.deg - 1,
                                             grad_phi_nodal(self.deg, self.K, self.gl[e], i), self.vol[e])
                load[self.DoF(m.elem_nodes[e, i])] += cstress @ gint
        return load

    # ---- assembly
    def per_element_stiffness(self):
        return per_element_stiffness_batch(self.deg, self.K, self.gl, self.vol, self.C4)

    def assembleStiffnessMatrix(self):
        """m_assembleStiffnessMatrix (:1408-1466): upper-triangle triplets of K (NOT summed)."""
        m, N = self.mesh, self.N
        Ke = self.per_element_stiffness()
        dof = self.dof_array()[m.elem_nodes]                          # [nE, n]
        n = m.nodes_per_elem
        gi = (N * dof[:, :, None] + np.arange(N)[None, None, :]).reshape(len(dof), n * N)
        I = np.broadcast_to(gi[:, :, None], Ke.shape)
        J = np.broadcast_to(gi[:, None, :], Ke.shape)
        di = np.repeat(dof, N, axis=1)
        keep = (di[:, :, None] <= di[:, None, :]) & (I <= J)          # :1421,1425
        nvar = N * self.numDoFs()
        return TripletMatrix.from_arrays(nvar, nvar, I[keep], J[keep], Ke[keep])

    def dirichlet_vars_and_values(self):                              # :1469-1518
        m = self.mesh
        cdofs, cidx = [], {}
        vals, masks = [], []
        for bn in m.bdry_nodes:
            if not self.dirichletMask[bn].any():
                continue
            d = self.DoF(bn)
            if d not in cidx:
                cidx[d] = len(cdofs)
                cdofs.append(d)
                vals.append(self.dirichletValue[bn].copy())
                masks.append(self.dirichletMask[bn].copy())
            else:
                k = cidx[d]
                if (np.linalg.norm(self.dirichletValue[bn] - vals[k]) > 1e-10) or (masks[k] != self.dirichletMask[bn]).any():
                    raise RuntimeError("Mismatched Dirichlet constraint on periodic DoF")
        fv, fx = [], []
        for d, v, mk in zip(cdofs, vals, masks):
            for c in range(self.N):
                if mk[c]:
                    fv.append(self.N * d + c)
                    fx.append(v[c])
        return fv, fx

    def pin_node_vars(self):                                          # :1595-1618
        m = self.mesh
        interior = np.flatnonzero(~m.is_bdry_node)
        node = int(interior[0]) if len(interior) else 0
        return [self.N * self.DoF(node) + d for d in range(self.N)], [0.0] * self.N

    def build_system(self):                                           # :1377-1404, :1201-1249
        K = self.assembleStiffnessMatrix()
        fv, fx = [], []
        if self.useRigidMotionConstraint and self.useNRTPin:
            fv, fx = self.pin_node_vars()
        dv, dx = self.dirichlet_vars_and_values()
        fv, fx = fv + dv, fx + dx
        sys = SPSDSystem(K)
        sys.fix_variables(fv, fx)
        self._system = sys
        self._fixed = (fv, fx)
        return sys

    def solve(self, f=None):                                          # :479-487, :657
        if self._system is None:
            self.build_system()
        if f is None:
            f = self.neumannLoad()
        x = self._system.solve(np.asarray(f).ravel())
        return self.dofToNodeField(x)

    def dofToNodeField(self, x):                                      # :664-677
        x = np.asarray(x).reshape(-1, self.N)
        return x[self.dof_array()]

    # ---- post-processing
    def applyStiffnessMatrix(self, u_nodes):                          # :801-823  (matrix-free K u)
        m, N = self.mesh, self.N
        Ke = self.per_element_stiffness()
        ue = u_nodes[m.elem_nodes].reshape(len(m.elems), -1)
        fe = np.einsum('eij,ej->ei', Ke, ue).reshape(len(m.elems), m.nodes_per_elem, N)
        out = np.zeros((self.numDoFs(), N))
        np.add.at(out, self.dof_array()[m.elem_nodes], fe)
        return out

    def averageStrainField(self, u_nodes):                            # :99-123, :528-549
        """per-element average of sym(sum_i u_i (x) grad phi_i); flattened Voigt [nE, flatLen]
        (tensor shear components, not engineering)."""
        m, N = self.mesh, self.N
        out = np.zeros((len(m.elems), flat_len(N)))
        for e in range(len(m.elems)):
            eps = np.zeros((N, N))
            for i in range(m.nodes_per_elem):
                g = interpolant_integrate(self.K, self.deg - 1,
                                          grad_phi_nodal(self.deg, self.K, self.gl[e], i), 1.0)
                ui = u_nodes[m.elem_nodes[e, i]]
                eps += 0.5 * (np.outer(ui, g) + np.outer(g, ui))
            out[e] = flatten_sym(N, eps)
        return out

    def strainField(self, u_nodes, stress=False):                     # :511-526, Element::strain :99-117
        """per-element strain (stress) interpolant: nodal values [nE, nInterp, flatLen] of degree deg-1."""
        m, N = self.mesh, self.N
        nint = 1 if self.deg == 1 else self.K + 1
        out = np.zeros((len(m.elems), nint, flat_len(N)))
        for e in range(len(m.elems)):
            S = np.zeros((nint, N, N))
            for i in range(m.nodes_per_elem):
                g = grad_phi_nodal(self.deg, self.K, self.gl[e], i)
                ui = u_nodes[m.elem_nodes[e, i]]
                for k in range(nint):
                    S[k] += 0.5 * (np.outer(ui, g[k]) + np.outer(g[k], ui))
            for k in range(nint):
                out[e, k] = self.elem_D(e).double_contract_flat(flatten_sym(N, S[k])) if stress else flatten_sym(N, S[k])
        return out

    def averageStressField(self, u_nodes):
        eps = self.averageStrainField(u_nodes)
        return np.stack([self.elem_D(e).double_contract_flat(eps[e]) for e in range(len(eps))])


# --------------------------------------------------------------------------------------
# Periodic homogenization                                  PeriodicHomogenization.hh:34-186
# --------------------------------------------------------------------------------------
def solve_cell_problems(sim: Simulator, eps=1e-7):                    # :34-54
    sim.applyPeriodicConditions(eps)
    sim.useRigidMotionConstraint = True
    sim.useNRTPin = True
    w = []
    for k in range(flat_len(sim.N)):
        rhs = sim.constantStrainLoad(-canonical_strain(sim.N, k))
        w.append(sim.solve(rhs))
    return w


def homogenized_elasticity_tensor(sim: Simulator, w):
    """Stress-like form  Eh.DRow(j) = 1/|Y| sum_e vol_e [E_e : avg strain(w_j) + E_e.DRow(j)]
    (PeriodicHomogenization.hh:72-103 homogenizedElasticityTensor). Returned as the flattened D
    matrix with column j = row j (major symmetry holds up to solver tolerance)."""
    N = sim.N
    fl = flat_len(N)
    Ch = np.zeros((fl, fl))
    mn, mx = sim.mesh.bounding_box()
    tot = float(np.prod(mx - mn))        # |Y| = periodic cell (bounding box) volume, not the material volume
    for j in range(fl):
        eps_w = sim.averageStrainField(w[j])
        ej = flatten_sym(N, canonical_strain(N, j))
        for e in range(len(eps_w)):
            Ch[:, j] += sim.vol[e] * sim.elem_D(e).double_contract_flat(ej + eps_w[e])
    return Ch / tot


def homogenized_elasticity_tensor_displacement_form(sim: Simulator, w, base_cell_volume=0.0):
    """homogenizedElasticityTensorDisplacementForm (PeriodicHomogenization.hh:146-186): boundary-integral
    form used by the Python binding (periodic_homogenization.cc:59); constant base tensor = element 0's."""
    N, fl = sim.N, flat_len(sim.N)
    mesh = sim.mesh
    if base_cell_volume == 0.0:
        mn, mx = mesh.bounding_box()
        base_cell_volume = float(np.prod(mx - mn))
    EBase = sim.elem_D(0)
    bvol, bnrm = mesh.bdry_elem_geometry()
    wts = integrated_shape_functions(mesh.deg, mesh.K - 1)
    Eh = np.zeros((fl, fl))
    for b in range(len(mesh.bdry_elem_nodes)):
        n = bnrm[b]
        for i in range(fl):
            w_be = w[i][mesh.bdry_elem_nodes[b]]                       # boundary-node displacements (:170-171)
            w_int = (wts[:, None] * w_be).sum(axis=0) * bvol[b]       # w_be.integrate(be->volume())
            nw = 0.5 * (np.outer(w_int, n) + np.outer(n, w_int))
            Eh[i, :] += EBase.double_contract_flat(flatten_sym(N, nw))
    Eh += EBase.D * float(np.sum(sim.vol))
    return Eh / base_cell_volume


# --------------------------------------------------------------------------------------
# Orthotropic-cell homogenization                         OrthotropicHomogenization.hh:44-219
# --------------------------------------------------------------------------------------
def face_membership(p, mn, mx, eps):
    """PeriodicBoundaryMatcher::FaceMembership (PeriodicBoundaryMatcher.hh:37-50): (onMin[d], onMax[d])."""
    p = np.asarray(p)
    return np.abs(p - mn) <= eps, np.abs(p - mx) <= eps


def ortho_cell_fixed_vars(sim: Simulator, eps=1e-7):
    """Fixed-variable sets of Orthotropic::solveCellProblems (OrthotropicHomogenization.hh:84-136):
    [stretch system, shear system s = 0 .. flatLen - N - 1]. Variables N * node + c, sorted."""
    m, N = sim.mesh, sim.N
    mn, mx = m.bounding_box()
    bnodes = np.unique(m.bdry_elem_nodes)
    pos = m.node_pos
    sets = []
    stretch = []
    for n in bnodes:                                                  # :88-94  w^ii_c = 0 on reflection plane c
        lo, hi = face_membership(pos[n], mn, mx, eps)
        for c in range(N):
            if lo[c] or hi[c]:
                stretch.append(N * n + c)
    sets.append(np.array(sorted(stretch), dtype=np.int64))
    for s_ in range(flat_len(N) - N):                                 # :105-136
        fix = np.zeros(N * m.num_nodes, dtype=bool)
        for n in bnodes:
            lo, hi = face_membership(pos[n], mn, mx, eps)
            for c in range(N):
                if lo[c] or hi[c]:
                    if N == 3:
                        fix[N * n + s_] = True                        # perpendicular to the shear plane
                        if c != s_:
                            fix[N * n + (N - (c + s_))] = True        # neither c nor s
                    else:
                        fix[N * n + (1 if c == 0 else 0)] = True
        sets.append(np.flatnonzero(fix).astype(np.int64))
    return sets


def solve_cell_problems_orthotropic(sim: Simulator, eps=1e-7):
    """Orthotropic::solveCellProblems: no periodicity / rigid-motion constraint; one SPD system for the N stretch probes,
    one per shear probe (:44-153)."""
    sim.dofForNode, sim.numDoFs_ = None, sim.mesh.num_nodes          # removePeriodicConditions (:874-879)
    sim.beInternal[:] = False
    sim.useRigidMotionConstraint = False
    sim._system = None
    N = sim.N
    K = sim.assembleStiffnessMatrix()
    K.sum_repeated()
    systems = []
    for fv in ortho_cell_fixed_vars(sim, eps):
        sysm = SPSDSystem(K)
        sysm.fix_variables(fv, np.zeros(len(fv)))
        systems.append(sysm)
    w = []
    for ij in range(flat_len(N)):
        l = sim.constantStrainLoad(-canonical_strain(N, ij))
        sysm = systems[0] if ij < N else systems[ij - N + 1]
        w.append(sysm.solve(l.ravel()).reshape(-1, N))
    return w


def fluctuation_displacement_sign(N, ij, r):
    """OrthotropicHomogenization.hh:161-174."""
    if ij < N:
        return 1.0
    bits = [(r >> b) & 1 for b in range(N)]
    if N == 3:
        bits[ij - N] = 0
    return -1.0 if sum(bits) == 1 else 1.0


def homogenized_tensor_from_ortho_cell_quantity(N, EhO):
    """OrthotropicHomogenization.hh:183-198 (upper triangle accumulated, major-symmetric result)."""
    fl = flat_len(N)
    Eh = np.zeros((fl, fl))
    for r in range(1 << N):
        for kl in range(fl):
            for ij in range(kl + 1):
                Eh[ij, kl] += fluctuation_displacement_sign(N, ij, r) * fluctuation_displacement_sign(N, kl, r) * EhO[ij, kl]
    Eh /= (1 << N)
    return np.triu(Eh) + np.triu(Eh, 1).T


# --------------------------------------------------------------------------------------
# Discrete shape derivatives (forward mode)     LinearElasticity.hh:234-330, :1297-1374;
# EmbeddedElement.hh:269-278,338-372; PeriodicHomogenization.hh:484-491,527-563
# The reference pins none of these with a test; they are anchored in tests/ on central finite differences of
# this oracle's own K(p), constantStrainLoad(p), strains and Ch(p) ("parity unpinned", exact mathematics).
# --------------------------------------------------------------------------------------
def delta_grad_barycentric(gl, i, delta_p):
    """EmbeddedElement.hh:269-278: delta grad lambda_i = - sum_k grad lambda_k (grad lambda_i . delta_p[k]).
    gl: N x (K+1) (column k = grad lambda_k); delta_p: (K+1) x N corner perturbations."""
    res = np.zeros(gl.shape[0])
    for k in range(gl.shape[1]):
        res -= gl[:, k] * np.dot(gl[:, i], delta_p[k])
    return res


def relative_delta_volume(gl, delta_p):
    """EmbeddedElement.hh:366-372."""
    return sum(np.dot(gl[:, k], delta_p[k]) for k in range(gl.shape[1]))


def delta_grad_phi_nodal(deg, K, gl, i, delta_p):
    """EmbeddedElement::deltaGradPhi (:338-363): nodal values of the change of the grad phi_i interpolant."""
    nv = K + 1
    if deg == 1:
        return delta_grad_barycentric(gl, i, delta_p)[None, :].copy()
    out = np.zeros((nv, gl.shape[0]))
    if i < nv:
        d = delta_grad_barycentric(gl, i, delta_p)
        for j in range(nv):
            out[j] = -d
        out[i] *= -3
    else:
        e = i - nv
        out[EDGE_START[e]] = 4 * delta_grad_barycentric(gl, EDGE_END[e], delta_p)
        out[EDGE_END[e]] = 4 * delta_grad_barycentric(gl, EDGE_START[e], delta_p)
    return out


def _vec_phi_strains(N, n, gphi_of):
    """vecPhiStrains / deltaVecPhiStrains (LinearElasticity.hh:79-97, :239-254): strain interpolant (nodal symmetric
    matrices [nInterp, N, N]) of the vector basis function i*N + c, from the nodal gradient values gphi_of(i)."""
    out = []
    for i in range(n):
        g = gphi_of(i)
        for c in range(N):
            S = np.zeros((g.shape[0], N, N))
            for inode in range(g.shape[0]):
                for var in range(N):
                    val = (1.0 if var == c else 0.5) * g[inode, var]
                    S[inode, c, var] += val                         # SymmetricMatrix (c, var): one stored entry
                    if var != c:
                        S[inode, var, c] += val
            out.append(S)
    return out


def delta_per_element_stiffness_loop(deg, K, gl, vol, D: "ElasticityTensor", delta_p):
    """Literal restatement of deltaPerElementStiffness (LinearElasticity.hh:306-330), UPPER triangle (NaN below)."""
    N = gl.shape[0]
    n = num_nodes(K, deg)
    strain_phi = _vec_phi_strains(N, n, lambda i: grad_phi_nodal(deg, K, gl, i))
    dstrain_phi = _vec_phi_strains(N, n, lambda i: delta_grad_phi_nodal(deg, K, gl, i, delta_p))
    dvol = vol * relative_delta_volume(gl, delta_p)
    stress_phi = [np.stack([D.double_contract(S[k]) for k in range(S.shape[0])]) for S in strain_phi]
    qdeg = 2 * (deg - 1)
    dKe = np.full((n * N, n * N), np.nan)
    at = lambda T, p: eval_interpolant(K, deg - 1, T, p)
    for i in range(n * N):
        for j in range(i, n * N):
            v = integrate(K, qdeg, lambda p: np.sum(at(stress_phi[i], p) * at(dstrain_phi[j], p))
                          + np.sum(at(stress_phi[j], p) * at(dstrain_phi[i], p)), vol)
            v += integrate(K, qdeg, lambda p: np.sum(at(stress_phi[i], p) * at(strain_phi[j], p)), dvol)
            dKe[i, j] = v
    return dKe


def delta_gl_batch(gl_batch, dp_batch):
    """Vectorised delta grad lambda and delta vol / vol: gl_batch [nE, N, nv], dp_batch [nE, nv, N]."""
    s = np.einsum('eai,eka->eik', gl_batch, dp_batch)                # grad lambda_i . delta_p_k
    dgl = -np.einsum('eak,eik->eai', gl_batch, s)
    rel = np.einsum('eak,eka->e', gl_batch, dp_batch)
    return dgl, rel


def delta_per_element_stiffness_batch(deg, K, gl_batch, vol_batch, C4_batch, dp_batch):
    """Full symmetric delta Ke for many elements: product rule on per_element_stiffness_batch."""
    dgl, rel = delta_gl_batch(gl_batch, dp_batch)
    G, w = gradphi_at_quadrature(deg, K, gl_batch)
    dG, _ = gradphi_at_quadrature(deg, K, dgl)
    nE, nq, n, N = G.shape
    H = np.einsum('q,eqia,eqjb->eiajb', w, dG, G, optimize=True) + np.einsum('q,eqia,eqjb->eiajb', w, G, dG, optimize=True) \
        + rel[:, None, None, None, None] * np.einsum('q,eqia,eqjb->eiajb', w, G, G, optimize=True)
    dKe = np.einsum('eiajb,eacdb->eicjd', H, np.broadcast_to(C4_batch, (nE, N, N, N, N)), optimize=True)
    return (dKe * vol_batch[:, None, None, None, None]).reshape(nE, n * N, n * N)


def _corner_perturbations(sim: "Simulator", deltaP):
    """extractElementCornerValues (LinearElasticity.hh:1290-1295): [nE, nv, N]; vertex v is node v."""
    m = sim.mesh
    return np.asarray(deltaP, dtype=np.float64)[m.elem_nodes[:, :m.K + 1]]


def apply_delta_stiffness_matrix(sim: "Simulator", u_nodes, deltaP):
    """Simulator::applyDeltaStiffnessMatrix (LinearElasticity.hh:1301-1328): per-node u -> per-DoF load."""
    m, N = sim.mesh, sim.N
    dKe = delta_per_element_stiffness_batch(sim.deg, sim.K, sim.gl, sim.vol, sim.C4, _corner_perturbations(sim, deltaP))
    ue = np.asarray(u_nodes)[m.elem_nodes].reshape(len(m.elems), -1)
    fe = np.einsum('eij,ej->ei', dKe, ue).reshape(len(m.elems), m.nodes_per_elem, N)
    out = np.zeros((sim.numDoFs(), N))
    np.add.at(out, sim.dof_array()[m.elem_nodes], fe)
    return out


def delta_constant_strain_load(sim: "Simulator", cstrain, deltaP):
    """Simulator::deltaConstantStrainLoad (:1331-1348) with deltaPerElementConstantStrainLoad (:289-304):
    l(c,i) = [int_vol delta strain(phi_ic) + int_dvol strain(phi_ic)] : (C : cstrain)."""
    m, N = sim.mesh, sim.N
    dp = _corner_perturbations(sim, deltaP)
    load = np.zeros((sim.numDoFs(), N))
    for e in range(len(m.elems)):
        gl, vol = sim.gl[e], sim.vol[e]
        s = sim.elem_D(e).double_contract(cstrain)
        dvol = vol * relative_delta_volume(gl, dp[e])
        phi = _vec_phi_strains(N, m.nodes_per_elem, lambda i: grad_phi_nodal(sim.deg, sim.K, gl, i))
        dphi = _vec_phi_strains(N, m.nodes_per_elem, lambda i: delta_grad_phi_nodal(sim.deg, sim.K, gl, i, dp[e]))
        for i in range(m.nodes_per_elem):
            for c in range(N):
                l = np.sum(interpolant_integrate(sim.K, sim.deg - 1, dphi[i * N + c], vol) * s)
                l += np.sum(interpolant_integrate(sim.K, sim.deg - 1, phi[i * N + c], dvol) * s)
                load[sim.DoF(m.elem_nodes[e, i]), c] += l
    return load


def delta_average_strain_field(sim: "Simulator", u_nodes, delta_u, deltaP):
    """Simulator::deltaAverageStrainField (:1364-1374): (delta strain)(u).average() + strain(delta u).average(),
    with deltaStrain (:259-277). Flattened [nE, flatLen]."""
    m, N = sim.mesh, sim.N
    dp = _corner_perturbations(sim, deltaP)
    out = sim.averageStrainField(np.asarray(delta_u))
    for e in range(len(m.elems)):
        deps = np.zeros((N, N))
        for i in range(m.nodes_per_elem):
            dg = interpolant_integrate(sim.K, sim.deg - 1, delta_grad_phi_nodal(sim.deg, sim.K, sim.gl[e], i, dp[e]), 1.0)
            ui = np.asarray(u_nodes)[m.elem_nodes[e, i]]
            deps += 0.5 * (np.outer(ui, dg) + np.outer(dg, ui))
        out[e] += flatten_sym(N, deps)
    return out


def delta_fluctuation_displacements(sim: "Simulator", w, deltaP):
    """deltaFluctuationDisplacements (PeriodicHomogenization.hh:527-544)."""
    out = []
    for ij in range(len(w)):
        rhs = delta_constant_strain_load(sim, -canonical_strain(sim.N, ij), deltaP)
        rhs -= apply_delta_stiffness_matrix(sim, w[ij], deltaP)
        out.append(sim.solve(rhs))
    return out


def _strain_at_quadrature(sim: "Simulator", u_nodes, gl_batch):
    """sym(sum_i u_i (x) grad phi_i) at the points of the degree-2(deg-1) rule: [nE, nq, N, N]."""
    G, w = gradphi_at_quadrature(sim.deg, sim.K, gl_batch)
    ue = np.asarray(u_nodes)[sim.mesh.elem_nodes]                     # [nE, n, N]
    Gu = np.einsum('eic,eqib->eqcb', ue, G)
    return 0.5 * (Gu + np.transpose(Gu, (0, 1, 3, 2))), w


def mutual_energies(sim: "Simulator", w, deltaP=None):
    """sum_e int (e^ij + eps(w^ij)) : C : (e^kl + eps(w^kl)) dV  (= |Y| Ch, energy form), or with deltaP the volume
    form of its discrete shape derivative quoted at PeriodicHomogenization.hh:484-491."""
    N, fl = sim.N, flat_len(sim.N)
    nE = len(sim.mesh.elems)
    C4 = np.broadcast_to(sim.C4, (nE,) + (N,) * 4)
    G, dG = [], []
    if deltaP is not None:
        dgl, rel = delta_gl_batch(sim.gl, _corner_perturbations(sim, deltaP))
    for ij in range(fl):
        S, wq = _strain_at_quadrature(sim, w[ij], sim.gl)
        G.append(S + canonical_strain(N, ij)[None, None])
        if deltaP is not None:
            dG.append(_strain_at_quadrature(sim, w[ij], dgl)[0])
    out = np.zeros((fl, fl))
    for ij in range(fl):
        for kl in range(fl):
            if deltaP is None:
                v = np.einsum('q,e,eqab,eabcd,eqcd->', wq, sim.vol, G[ij], C4, G[kl], optimize=True)
            else:
                v = np.einsum('q,e,eqab,eabcd,eqcd->', wq, sim.vol * rel, G[ij], C4, G[kl], optimize=True) \
                    + np.einsum('q,e,eqab,eabcd,eqcd->', wq, sim.vol, dG[ij], C4, G[kl], optimize=True) \
                    + np.einsum('q,e,eqab,eabcd,eqcd->', wq, sim.vol, G[ij], C4, dG[kl], optimize=True)
            out[ij, kl] = v
    return out


def boundary_strain_field(sim: "Simulator", u_nodes, stress=False):
    """restrictInterpolant (InterpolantRestriction.hh:29-66) of Element::strain / stress to every boundary element:
    [nBE, 1 | N, flatLen], the parent's interpolant at the boundary element's corners in the boundary vertex order."""
    m, N = sim.mesh, sim.N
    vol = sim.strainField(u_nodes, stress=stress)
    nb = len(m.bdry_elem_verts)
    out = np.zeros((nb, 1 if sim.deg == 1 else N, flat_len(N)))
    for b in range(nb):
        e = int(m.bdry_parent[b])
        if sim.deg == 1:
            out[b, 0] = vol[e, 0]                                       # degree 0 interpolants are not nodal (:45-48)
            continue
        for sdni in range(N):                                           # brute-force node search (:52-63)
            for dni in range(sim.K + 1):
                if m.elem_nodes[e, dni] == m.bdry_elem_verts[b, sdni]:
                    out[b, sdni] = vol[e, dni]
    return out


def _simplex_monomial_integral(d, alpha):
    """int over the unit-volume d-simplex of prod lambda_i^alpha_i = d! prod alpha_i! / (|alpha| + d)!"""
    from math import factorial
    num = factorial(d)
    for a in alpha:
        num *= factorial(a)
    return num / factorial(sum(alpha) + d)


def homogenized_elasticity_tensor_gradient(sim: "Simulator", w):
    """homogenizedElasticityTensorGradient (PeriodicHomogenization.hh:226-288): per boundary element the nodal values of
    G_ijkl = 1/|bbox| (e_ij + eps(w_ij)) : E : (e_kl + eps(w_kl)), a degree 2 (Deg - 1) interpolant; zero on the periodic
    (internal) boundary. [nBE, nNodes(GDeg), flatLen, flatLen] (major-symmetric storage: upper triangle mirrored)."""
    m, N, K, deg = sim.mesh, sim.N, sim.K, sim.deg
    fl = flat_len(N)
    bbox_vol = float(np.prod(m.node_pos.max(axis=0) - m.node_pos.min(axis=0)))      # mesh.boundingBox().volume() (:237)
    nb = len(m.bdry_elem_nodes)
    nn = 1 if deg == 1 else m.bdry_elem_nodes.shape[1]
    strains = [sim.strainField(w[ij]) for ij in range(fl)]
    out = np.zeros((nb, nn, fl, fl))
    for b in range(nb):
        if sim.beInternal[b]:
            continue
        e = int(m.bdry_parent[b])
        en = list(m.elem_nodes[e])
        D = sim.elem_D(e)
        for n in range(nn):
            def strain_at(ij):
                if deg == 1:
                    s = strains[ij][e, 0]
                else:
                    dni = en.index(m.bdry_elem_nodes[b, n])             # restriction by node id (InterpolantRestriction.hh:52-63)
                    if dni <= K:
                        s = strains[ij][e, dni]
                    else:                                               # linear interpolant at an edge midpoint
                        s = 0.5 * (strains[ij][e, EDGE_START[dni - K - 1]] + strains[ij][e, EDGE_END[dni - K - 1]])
                return s + flatten_sym(N, canonical_strain(N, ij))
            for ij in range(fl):
                sij = D.double_contract_flat(strain_at(ij))
                for kl in range(ij, fl):
                    g = strain_at(kl)
                    v = sum(sij[c] * g[c] * (1.0 if c < N else 2.0) for c in range(fl)) / bbox_vol
                    out[b, n, ij, kl] = out[b, n, kl, ij] = v
    return out


def delta_homogenized_elasticity_tensor_boundary_form(sim: "Simulator", w, delta_p):
    """deltaHomogenizedElasticityTensor (PeriodicHomogenization.hh:492-514): the boundary integral of the linear normal
    velocity n . delta_p against homogenizedElasticityTensorGradient, integrated exactly."""
    m, N, deg = sim.mesh, sim.N, sim.deg
    fl = flat_len(N)
    sd = homogenized_elasticity_tensor_gradient(sim, w)
    bvol, bnrm = m.bdry_elem_geometry()
    d = N - 1                                                           # boundary simplex dimension
    out = np.zeros((fl, fl))
    for b in range(len(m.bdry_elem_verts)):
        nsv = [float(np.dot(bnrm[b], delta_p[m.bdry_elem_verts[b, a]])) for a in range(N)]
        for a in range(N):
            for n in range(sd.shape[1]):
                if deg == 1:
                    wgt = 1.0 / N                                       # int lambda_a
                elif n < N:                                             # phi_n = lambda_n (2 lambda_n - 1)
                    al2 = [0] * N; al2[n] += 2; al2[a] += 1
                    al1 = [0] * N; al1[n] += 1; al1[a] += 1
                    wgt = 2 * _simplex_monomial_integral(d, al2) - _simplex_monomial_integral(d, al1)
                else:                                                   # phi = 4 lambda_s lambda_t on boundary edge n - N
                    al = [0] * N; al[EDGE_START[n - N]] += 1; al[EDGE_END[n - N]] += 1; al[a] += 1
                    wgt = 4 * _simplex_monomial_integral(d, al)
                out += bvol[b] * wgt * nsv[a] * sd[b, n]
    return out


def change_in_div_tensor_load(sim: "Simulator", vn, t, ignore_periodic_bdry=True):
    """Simulator::changeInDivTensorLoad (LinearElasticity.hh:590-650), literal loops: the per-DoF load
    -int_bdry vn strain(phi_n e_c) : t dA for the boundary tensor interpolants t [nBE, 1 | N, flatLen] (degree Deg - 1)
    and the linear normal velocity vn [nBE, N] (boundary vertex order). strain(phi e_c) : t = (t grad phi)_c."""
    m, N, K, deg = sim.mesh, sim.N, sim.K, sim.deg
    load = np.zeros((sim.numDoFs(), N))
    dof = sim.dof_array()
    bvol, _ = m.bdry_elem_geometry()
    pts, wts = quadrature_rule(K - 1, 1 + 2 * (deg - 1))                 # IntegrandDeg (:628-629)
    for b in range(len(m.bdry_elem_verts)):
        if ignore_periodic_bdry and sim.beInternal[b]:
            continue
        e = int(m.bdry_parent[b])
        q = [list(m.elem_nodes[e, :K + 1]).index(v) for v in m.bdry_elem_verts[b]]
        for mu, wq in zip(pts, wts):
            lam = np.zeros(K + 1)
            lam[q] = mu
            gphi = grad_phis_at(deg, K, sim.gl[e], lam)                  # restrictInterpolant of the volume phi strains
            tv = unflatten_sym(N, t[b, 0] if deg == 1 else sum(mu[c] * t[b, c] for c in range(N)))
            vnv = float(np.dot(mu, vn[b]))
            for n in range(m.nodes_per_elem):
                load[dof[m.elem_nodes[e, n]]] -= bvol[b] * wq * vnv * (tv @ gphi[:, n])
    return load


def fluctuation_displacement_shape_derivatives(sim: "Simulator", w, vn, project_out_normal_stress=False):
    """fluctuationDisplacementShapeDerivatives (PeriodicHomogenization.hh:301-370): the Eulerian shape derivative of every
    w^kl under the normal boundary velocity vn, from cell problems loaded with -int_bdry vn strain(phi) : C : [strain(w^kl) + e^kl]."""
    m, N = sim.mesh, sim.N
    _, bnrm = m.bdry_elem_geometry()
    out = []
    for kl in range(len(w)):
        st = boundary_strain_field(sim, w[kl]) + flatten_sym(N, canonical_strain(N, kl))[None, None, :]
        for b in range(len(st)):
            D = sim.elem_D(int(m.bdry_parent[b]))
            for n in range(st.shape[1]):
                st[b, n] = 0.0 if sim.beInternal[b] else D.double_contract_flat(st[b, n])
                if project_out_normal_stress and not sim.beInternal[b]:   # s - (sn) n^T - n (sn)^T + (n^T s n) n n^T (:343-355)
                    nn = bnrm[b]
                    sm = unflatten_sym(N, st[b, n])
                    half = 0.5 * (np.outer(sm @ nn, nn) + np.outer(nn, sm @ nn))
                    sm = sm - 2.0 * half
                    sm = sm - float(nn @ sm @ nn) * np.outer(nn, nn)
                    st[b, n] = flatten_sym(N, sm)
        out.append(sim.solve(change_in_div_tensor_load(sim, vn, st, True)))
    return out


def homogenized_elasticity_tensor_discrete_differential(sim: "Simulator", w, base_cell_volume=0.0):
    """homogenizedElasticityTensorDiscreteDifferential (PeriodicHomogenization.hh:372-480), literal loops: the one-form
    dCh[vertex, component] as flatLen x flatLen tensors (upper triangle ij <= kl filled, mirrored at the end).
    Returns [nVert, N, flatLen, flatLen]."""
    m, N, K, deg = sim.mesh, sim.N, sim.K, sim.deg
    fl = flat_len(N)
    nv = K + 1
    nvert = int(m.elem_nodes[:, :nv].max()) + 1
    out = np.zeros((nvert, N, fl, fl))
    at = lambda T, p: eval_interpolant(K, deg - 1, T, p)
    for e in range(len(m.elems)):
        gl, vol, D = sim.gl[e], sim.vol[e], sim.elem_D(e)
        nodes = m.elem_nodes[e]
        gphi = [grad_phi_nodal(deg, K, gl, n) for n in range(m.nodes_per_elem)]            # :429-431
        ninterp = gphi[0].shape[0]
        strain, stress = [], []
        for ij in range(fl):                                                               # :421-426
            S = np.zeros((ninterp, N, N))
            for n in range(m.nodes_per_elem):
                for k in range(ninterp):
                    S[k] += 0.5 * (np.outer(w[ij][nodes[n]], gphi[n][k]) + np.outer(gphi[n][k], w[ij][nodes[n]]))
            S += canonical_strain(N, ij)[None]
            strain.append(S)
            stress.append(np.stack([D.double_contract(S[k]) for k in range(ninterp)]))
        for ij in range(fl):
            for kl in range(ij, fl):
                mutual = integrate(K, 2 * (deg - 1), lambda p: np.sum(at(strain[ij], p) * at(stress[kl], p)), vol)   # :437-438
                for v in range(nv):
                    out[nodes[v], :, ij, kl] += mutual * gl[:, v]                         # :439-442
                for n in range(m.nodes_per_elem):
                    scw = np.stack([stress[kl][k] @ w[ij][nodes[n]] for k in range(ninterp)])                         # :449-454
                    if ij != kl:
                        scw = scw + np.stack([stress[ij][k] @ w[kl][nodes[n]] for k in range(ninterp)])
                    else:
                        scw = 2 * scw
                    for v in range(nv):
                        gb = gl[:, v]
                        term = integrate(K, deg, lambda p: np.dot(gb, at(scw, p)) * at(gphi[n], p), vol)             # :459-461
                        out[nodes[v], :, ij, kl] -= term
    if base_cell_volume == 0.0:
        mn, mx = m.bounding_box()
        base_cell_volume = float(np.prod(mx - mn))
    out /= base_cell_volume                                                                # :477
    iu = np.triu_indices(fl, 1)
    out[:, :, iu[1], iu[0]] = out[:, :, iu[0], iu[1]]
    return out


# --------------------------------------------------------------------------------------
# Scalar operators on the same mesh: Laplacian.hh, MassMatrix.hh, Poisson.hh
# --------------------------------------------------------------------------------------
def laplacian_triplets(mesh: FEMMesh):
    """Laplacian::construct (Laplacian.hh:27-57,97-104): upper triangle of int grad phi_i . grad phi_j with
    Quadrature<N, 2(Deg-1)>, local loop i <= j, entry placed at (min, max) of the global node indices.
    The degree-1 specialisation (:60-81) is the same formula with constant gradients."""
    K, deg = mesh.K, mesh.deg
    vol, gl = mesh.embeddings_batch()
    n = mesh.nodes_per_elem
    pts, w = quadrature_rule(K, 2 * (deg - 1))
    T = TripletMatrix(mesh.num_nodes, mesh.num_nodes)
    I, J, V = [], [], []
    for e in range(len(mesh.elems)):
        G = [grad_phis_at(deg, K, gl[e], p) for p in pts]           # each N x n
        nodes = mesh.elem_nodes[e]
        for i in range(n):
            for j in range(i, n):
                val = 0.0
                for q in range(len(w)):
                    val += w[q] * float(G[q][:, i] @ G[q][:, j])
                val *= vol[e]
                ni, nj = int(nodes[i]), int(nodes[j])
                I.append(min(ni, nj)); J.append(max(ni, nj)); V.append(val)
    return TripletMatrix.from_arrays(mesh.num_nodes, mesh.num_nodes, np.array(I), np.array(J), np.array(V))


def mass_triplets(mesh: FEMMesh, lumped=False):
    """MassMatrix::construct (MassMatrix.hh:50-86,103-128): upper triangle of int phi_i phi_j with
    Quadrature<K, 2 Deg>; `lumped` sums every row of the full matrix onto the diagonal (:110-124)."""
    K, deg = mesh.K, mesh.deg
    vol, _ = mesh.embeddings_batch()
    n = mesh.nodes_per_elem
    pts, w = quadrature_rule(K, 2 * deg)
    Phi = np.array([shape_functions(deg, K, p) for p in pts])      # nq x n
    Mref = np.einsum('q,qi,qj->ij', w, Phi, Phi)
    I, J, V = [], [], []
    for e in range(len(mesh.elems)):
        nodes = mesh.elem_nodes[e]
        for i in range(n):
            for j in range(n):
                ni, nj = int(nodes[i]), int(nodes[j])
                if nj < ni:
                    continue                                         # upper triangle only (:63)
                I.append(ni); J.append(nj); V.append(Mref[i, j] * vol[e])
    I, J, V = np.array(I), np.array(J), np.array(V)
    if lumped:
        diag = np.zeros(mesh.num_nodes)
        np.add.at(diag, I, V)
        off = I != J
        np.add.at(diag, J[off], V[off])
        r = np.arange(mesh.num_nodes)
        return TripletMatrix.from_arrays(mesh.num_nodes, mesh.num_nodes, r, r, diag)
    return TripletMatrix.from_arrays(mesh.num_nodes, mesh.num_nodes, I, J, V)


def poisson_solve(mesh: FEMMesh, dirichlet_boxes):
    """PoissonMesh::applyBoundaryConditions + solve (Poisson.hh:57-117): Dirichlet value on the BOUNDARY
    nodes inside each (inclusive) box -- later conditions overwrite earlier ones -- zero right-hand side,
    zero-Neumann elsewhere. dirichlet_boxes: list of (min_corner, max_corner, value)."""
    ctype = {}
    for mn, mx, val in dirichlet_boxes:
        mn, mx = np.asarray(mn, dtype=np.float64), np.asarray(mx, dtype=np.float64)
        for bn in mesh.bdry_nodes:
            p = mesh.node_pos[bn]
            if np.all(p >= mn) and np.all(p <= mx):
                ctype[int(bn)] = float(val)
    L = laplacian_triplets(mesh)
    system = SPSDSystem(L)
    fixed = [bn for bn in mesh.bdry_nodes if int(bn) in ctype]      # boundary-node order (:106-112)
    system.fix_variables([int(b) for b in fixed], [ctype[int(b)] for b in fixed])
    return system.solve(np.zeros(mesh.num_nodes)), np.array(fixed, dtype=np.int64)


def grad_u_average(mesh: FEMMesh, u):
    """PoissonMesh::gradUAverage (Poisson.hh:121-131): average over the element of sum_i u_i grad phi_i
    (the degree-(Deg-1) interpolant's average = mean of its vertex values, Functions.hh:246-253)."""
    K, deg = mesh.K, mesh.deg
    _, gl = mesh.embeddings_batch()
    out = np.zeros((len(mesh.elems), mesh.N))
    for e in range(len(mesh.elems)):
        nodes = mesh.elem_nodes[e]
        acc = np.zeros(mesh.N)
        for i in range(mesh.nodes_per_elem):
            gp = grad_phi_nodal(deg, K, gl[e], i)                    # nodal values of the interpolant
            acc += u[nodes[i]] * gp.mean(axis=0)
        out[e] = acc
    return out


# --------------------------------------------------------------------------------------
# Constrained (KKT) branch of Simulator::solve       LinearElasticity.hh:1169-1249,1520-1618
# --------------------------------------------------------------------------------------
def rotation_rows(sim: Simulator):
    """m_appendInfinitesimalRotationMatrix (:1525-1566); empty under periodic conditions (:1534-1542)."""
    N, m = sim.N, sim.mesh
    nd = sim.numDoFs()
    if (N == 2 and nd < m.num_nodes) or nd < m.num_nodes - 1:
        return np.zeros((0, N * nd))
    if nd < m.num_nodes:
        raise RuntimeError("Single pair periodic BC unsupported in 3D.")
    x = m.node_pos
    if N == 3:
        R = np.zeros((3, 3 * nd))
        R[0, 1::3], R[0, 2::3] = -x[:, 2], x[:, 1]
        R[1, 0::3], R[1, 2::3] = x[:, 2], -x[:, 0]
        R[2, 0::3], R[2, 1::3] = -x[:, 1], x[:, 0]
        return R
    R = np.zeros((1, 2 * nd))
    R[0, 0::2], R[0, 1::2] = -x[:, 1], x[:, 0]
    return R


def translation_rows(sim: Simulator, comps):
    """m_appendTranslationMatrix (:1568-1590): one row per requested component, ones on every DoF."""
    N, nd = sim.N, sim.numDoFs()
    T = np.zeros((len(comps), N * nd))
    for r, c in enumerate(comps):
        T[r, c::N] = 1.0
    return T


def solve_constrained(sim: Simulator, f=None, use_pin=False, no_rigid_motion=False, allow_ill_posed=False, rm_rhs=None):
    """Simulator::solve through assembleConstrainedSystem (:1201-1249) and SPSDSystem::setConstrained / solve
    (SparseMatrices.hh:2340-2348,2572-2590): the KKT system [[K, C^T], [C, 0]] on the free variables is solved
    with a general sparse LU (scipy splu standing in for UMFPACK)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    N, m = sim.N, sim.mesh
    n = N * sim.numDoFs()
    fv, fx = [], []
    rows = []
    crhs = np.zeros(0)

    def pin(comps):
        interior = np.flatnonzero(~m.is_bdry_node)
        node = int(interior[0]) if len(interior) else 0
        for c in comps:
            fv.append(N * sim.DoF(node) + c); fx.append(0.0)

    if no_rigid_motion:
        rows.append(rotation_rows(sim))
        if use_pin:
            pin(range(N))
        else:
            rows.append(translation_rows(sim, list(range(N))))
        k = sum(len(r) for r in rows)
        crhs = np.zeros(k) if rm_rhs is None else np.asarray(rm_rhs, dtype=np.float64)
        if len(crhs) != k:
            raise RuntimeError("Invalid rigid motion RHS")
    elif not allow_ill_posed:
        needs_t = [True] * N
        total = 0
        for bn in m.bdry_nodes:
            for c in range(N):
                if sim.dirichletMask[bn, c]:
                    needs_t[c] = False
                    total += 1
        comps = [c for c in range(N) if needs_t[c]]
        if comps:
            if use_pin:
                pin(comps)
            else:
                rows.append(translation_rows(sim, comps))
                crhs = np.zeros(len(comps))
        if total == 0:
            raise RuntimeError("Unimplemented")
    dv, dx = sim.dirichlet_vars_and_values()
    fv, fx = np.array(fv + dv, dtype=np.int64), np.array(fx + dx, dtype=np.float64)
    C = np.vstack(rows) if rows else np.zeros((0, n))
    K = sim.assembleStiffnessMatrix().sum_repeated().to_scipy_full_from_upper().tocsr()
    if f is None:
        f = sim.neumannLoad()
    f = np.asarray(f, dtype=np.float64).ravel()
    free = np.ones(n, bool)
    free[fv] = False
    ubar = np.zeros(n)
    ubar[fv] = fx
    b = (f - K @ ubar)[free]
    Cf = C[:, free]
    c2 = crhs - C @ ubar
    k = len(C)
    A = sp.bmat([[K[free][:, free], sp.csr_matrix(Cf.T)], [sp.csr_matrix(Cf), None if k == 0 else sp.csr_matrix((k, k))]]).tocsc() if k else K[free][:, free].tocsc()
    sol = spla.splu(A).solve(np.concatenate([b, c2]))
    u = ubar.copy()
    u[free] = sol[:free.sum()]
    return sim.dofToNodeField(u)
