"""Scalar operators on the GPU path: mirrors of `Laplacian::construct`
(Laplacian.hh:97-104), `MassMatrix::construct` (MassMatrix.hh:103-128) and `PoissonMesh`
(Poisson.hh:55-132). They run on the kernels of the elasticity path with 1x1 blocks
(`Context.set_operator`): same mesh topology, pattern, gather lists, assembly kernel, SpMV and PCG.
Only the full-degree operators are provided (the reference's forced-degree-1 variants on a
quadratic mesh are not)."""
import numpy as np

from . import _lib as L
from .core import Context


def _context_for(elements, vertices, degree, device, op):
    c = Context(device)
    c.mesh_build(np.asarray(elements), np.asarray(vertices, dtype=np.float64), degree)
    c.set_operator(op)
    return c


class _Triplets:
    """Upper-triangle triplets after sumRepeated (column-major order, like TripletMatrix::dumpBinary)."""

    def __init__(self, n, i, j, v):
        self.m = self.n = n
        self.i, self.j, self.v = i, j, v

    @property
    def nnz(self):
        return len(self.v)

    def toSciPy(self, full=True):
        import scipy.sparse as sp
        A = sp.coo_matrix((self.v, (self.i.astype(np.int64), self.j.astype(np.int64))), shape=(self.m, self.n)).tocsr()
        if full:
            A = A + sp.triu(A, 1).T
        return A


def laplacian(elements, vertices, degree=1, device=0, ctx=None):
    """== Laplacian::construct: upper triangle of the (positive semi-definite) FEM Laplacian."""
    c = ctx or _context_for(elements, vertices, degree, device, L.OP_LAPLACIAN)
    c.set_operator(L.OP_LAPLACIAN)
    c.assemble()
    i, j, v = c.export_upper_triplets()
    return _Triplets(c.n_dof, i, j, v)


def mass_matrix(elements, vertices, degree=1, lumped=False, device=0, ctx=None):
    """== MassMatrix::construct: upper triangle of the mass matrix; `lumped` puts the row sums of the
    full matrix on the diagonal (MassMatrix.hh:110-124)."""
    c = ctx or _context_for(elements, vertices, degree, device, L.OP_MASS)
    c.set_operator(L.OP_MASS)
    c.assemble()
    if lumped:
        diag = c.apply_K(np.ones(c.n_dof))                 # row sums of the full symmetric matrix = M 1, on the device
        r = np.arange(c.n_dof, dtype=np.uint64)
        return _Triplets(c.n_dof, r, r.copy(), diag)
    i, j, v = c.export_upper_triplets()
    return _Triplets(c.n_dof, i, j, v)


class PoissonMesh:
    """== PoissonMesh<K, Deg, EmbeddingSpace>: -laplace u = 0 with Dirichlet values on boundary regions and the
    natural zero-Neumann condition elsewhere."""

    def __init__(self, elements, vertices, degree=1, device=0):
        self.ctx = _context_for(elements, vertices, degree, device, L.OP_LAPLACIAN)
        self.rtol, self.maxit = 1e-10, 100000
        self.info = None

    def numNodes(self):
        return self.ctx.n_node

    def nodes(self):
        return self.ctx.node_positions()

    def applyDirichletBox(self, min_corner, max_corner, value, relative=False):
        """One DirichletCondition of `applyBoundaryConditions` (Poisson.hh:69-84): boundary nodes inside the
        inclusive box get `value` (the reference encodes it as the displacement's first component)."""
        d = self.ctx.dim
        self.ctx.bc_dirichlet_box(min_corner, max_corner, [float(value)] + [0.0] * (d - 1), relative=relative,
                                  components=[True] + [False] * (d - 1))

    def solve(self):
        u = self.ctx.sim_solve(None, use_pin=False, rtol=self.rtol, maxit=self.maxit)
        self.info = self.ctx.last_info
        return u[:, 0]

    def gradUAverage(self, u):
        return self.ctx.average_gradient(u)
