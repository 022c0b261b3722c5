"""`differential_operators` (src/python_bindings/differential_operators.cc:21-90): laplacian, mass, mass_elasticity,
bilaplacian, gradient on the MI355X path -- the scalar operators of SURVEY 8 f3 (`mfh_set_operator`: the elasticity
path's assembly kernel with 1x1 blocks), returned as `sparse_matrices.TripletMatrix` like the binding.

`forceP1=True` on a quadratic mesh builds the degree-1 operator over the VERTICES (Laplacian::construct<1>,
MassMatrix::construct<1> with NodeGetter<1>, MassMatrix.hh:39-47): a degree-1 context of the same vertices / elements."""
import numpy as np

from meshfem_amd import _lib as L
from meshfem_amd.core import Context
from sparse_matrices import TripletMatrix


def _assembled(mesh, op, forceP1, device):
    c = Context(device)
    c.mesh_build(mesh.elements(), mesh.vertices(), 1 if forceP1 else mesh.degree)
    c.set_operator(op)
    c.assemble()
    return c


def _triplets(n, i, j, v, upperTriOnly):
    M = TripletMatrix.fromArrays(n, n, i, j, v, symmetry_mode="UPPER_TRIANGLE")
    if not upperTriOnly:
        M.reflectUpperTriangle()
    return M


def laplacian(mesh, forceP1=False, upperTriOnly=False, device=0):
    """Laplacian::construct (Laplacian.hh:97-104): int grad phi_i . grad phi_j (positive semi-definite sign)."""
    c = _assembled(mesh, L.OP_LAPLACIAN, forceP1, device)
    i, j, v = c.export_upper_triplets()
    M = _triplets(c.n_dof, i, j, v, upperTriOnly)
    c.close()
    return M


def mass(mesh, lumped=False, forceP1=False, upperTriOnly=False, device=0):
    """MassMatrix::construct (MassMatrix.hh:103-128); lumped: row sums of the full matrix on the diagonal."""
    c = _assembled(mesh, L.OP_MASS, forceP1, device)
    n = c.n_dof
    if lumped:
        r = np.arange(n, dtype=np.uint64)
        M = TripletMatrix.fromArrays(n, n, r, r.copy(), c.apply_K(np.ones(n)))       # M 1 on the device
    else:
        i, j, v = c.export_upper_triplets()
        M = _triplets(n, i, j, v, upperTriOnly)
    c.close()
    return M


def mass_elasticity(mesh, lumped=False, forceP1=False, upperTriOnly=False, device=0):
    """MassMatrix::construct_vector_valued (MassMatrix.hh:130-150): the scalar mass matrix on every component of the
    interleaved unknowns (x0, y0, ..., x1, ...)."""
    Ms = mass(mesh, lumped, forceP1, True, device)
    N = mesh.embeddingDimension
    i, j, v = Ms.arrays()
    i, j = i.astype(np.uint64), j.astype(np.uint64)
    I = np.concatenate([N * i + c for c in range(N)])
    J = np.concatenate([N * j + c for c in range(N)])
    M = TripletMatrix.fromArrays(N * Ms.m, N * Ms.n, I, J, np.tile(v, N), symmetry_mode="NONE" if lumped else "UPPER_TRIANGLE")
    if not upperTriOnly and not lumped:
        M.reflectUpperTriangle()
    return M


def bilaplacian(mesh, forceP1=False, device=0):
    """L M_lumped^-1 L as a scipy CSC matrix (differential_operators.cc:48-68)."""
    import scipy.sparse as sp
    Lm = laplacian(mesh, forceP1, False, device).toSciPy()
    d = mass(mesh, True, forceP1, True, device).toSciPy().diagonal()
    return (Lm @ sp.diags(1.0 / d) @ Lm).tocsc()


def gradient(mesh, scalarField, device=0):
    """Per-element gradient of a nodal scalar field, degree-1 meshes only like the binding (:70-79)."""
    if mesh.degree > 1:
        raise RuntimeError("Interpolant type bindings unimplemented...")
    u = np.asarray(scalarField, dtype=np.float64)
    if u.shape != (mesh.numNodes(),):
        raise RuntimeError("Incorrect scalar field size")
    c = Context(device)
    c.mesh_build(mesh.elements(), mesh.vertices(), 1)
    c.set_operator(L.OP_LAPLACIAN)
    g = c.average_gradient(u)
    c.close()
    return g
