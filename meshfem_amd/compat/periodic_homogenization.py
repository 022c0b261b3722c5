"""`periodic_homogenization` (src/python_bindings/periodic_homogenization.cc:36-183): `homogenize` and `probe`
on the MI355X path: cell problems by PCG on the device, homogenized tensor in the DISPLACEMENT form
(PeriodicHomogenization.hh:146-186) like the binding, fluctuations centred by default."""
import numpy as np

from meshfem_amd import _lib as L
from meshfem_amd import homogenization as H
from meshfem_amd.linear_elasticity import Simulator


class HomogenizationResult:
    def __init__(self, Ch, w_ij, strain_w_ij):
        self.Ch, self.w_ij, self.strain_w_ij = Ch, w_ij, strain_w_ij


def _tensor_like(Cbase, D):
    out = type(Cbase)() if hasattr(Cbase, "setIsotropic") else None
    if out is None:
        return D
    out.D = D
    return out


def homogenize(mesh, Cbase, orthotropicCell=False, manualPeriodicVerticesFile="", centerFluctuationDisplacements=True,
               ignorePeriodicMismatch=False, device=0, preconditioner=L.PRECOND_MULTIGRID, rtol=1e-10):
    N = mesh.embeddingDimension
    sim = Simulator(mesh.elements(), mesh.vertices(), mesh.degree, device)
    sim.rtol = rtol
    sim.ctx.set_preconditioner(preconditioner)
    sim.setMaterial(Cbase)
    if orthotropicCell:                                    # periodic_homogenization.cc:53-56
        Ch, w, _ = H.homogenize_orthotropic_cell(sim)
    else:
        w, _ = H.solve_cell_problems(sim, ignore_periodic_mismatch=ignorePeriodicMismatch,
                                     manual_periodic_vertices_file=manualPeriodicVerticesFile)
        Ch = H.homogenized_elasticity_tensor_displacement_form(sim, w)
    if centerFluctuationDisplacements:                     # periodic_homogenization.cc:62-70
        w = [x - x.mean(axis=0) for x in w]
    strain = [sim.averageStrainField(x) for x in w]
    return HomogenizationResult(_tensor_like(Cbase, Ch), w, strain)


def probe(mesh, arg, macroStrain, *rest, **kw):
    """probe(mesh, homogenizationResult, macroStrain) or probe(mesh, Cbase, macroStrain, ...)
    -> (u [numNodes x N], strain_u [numElements x flatLen])  (periodic_homogenization.cc:92-150)."""
    if isinstance(arg, HomogenizationResult):
        hr = arg
    else:
        hr = homogenize(mesh, arg, *rest, centerFluctuationDisplacements=False, **kw)
    N = mesh.embeddingDimension
    ms = np.asarray(getattr(macroStrain, "flat", macroStrain), dtype=np.float64)
    w = np.zeros_like(hr.w_ij[0])
    sw = np.zeros_like(hr.strain_w_ij[0])
    for i in range(len(hr.w_ij)):
        dbl = 1.0 if i < N else 2.0                        # shearDoubler
        w += dbl * ms[i] * hr.w_ij[i]
        sw += dbl * ms[i] * hr.strain_w_ij[i]
    nodes = mesh.nodes()
    lo = mesh.bbox[0]
    bn = mesh.boundaryNodes()
    for d in range(N):                                     # face-average translation removal (:114-130)
        on = bn[np.abs(nodes[bn, d] - lo[d]) < 1e-9]
        w[:, d] -= w[on, d].mean()
    E = H._unflatten(N, ms)
    u = w + nodes @ E.T
    return u, sw + ms[None, :]
