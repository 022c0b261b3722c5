"""`mesh` (src/python_bindings/mesh.cc:42-134,258-330): `Mesh(path | V, F, degree)` and `PeriodicCondition`.
Node numbering, boundary extraction and periodic matching are the library's host-side restatements of
FEMMesh.inl / TetMesh.inl / PeriodicBoundaryMatcher.hh (bit-identical to the oracle: tests/test_host_logic.py)."""
import numpy as np

from meshfem_amd import mesh_io
from meshfem_amd.core import Context


class _Mesh:
    def __init__(self, V, F, degree=1, embeddingDimension=3):
        V, F = np.asarray(V, dtype=np.float64), np.asarray(F, dtype=np.int64)
        K = F.shape[1] - 1
        N = K if (K == 3 or embeddingDimension == 2) else V.shape[1]
        if K == 2 and N == 3 and np.abs(V[:, 2]).max() == 0 and embeddingDimension != 3:
            N = 2
        if N not in (2, 3) or N != K:
            raise RuntimeError("only tet meshes in 3D and triangle meshes in 2D are on the GPU path")
        self._V, self._F, self._deg, self._K, self._N = np.ascontiguousarray(V[:, :N]), F, int(degree), K, N
        self._h = Context(-1)                                 # host-only context: topology, no device
        self._h.mesh_build(self._F, self._V, self._deg)

    # ---- geometry / connectivity (mesh.cc:48-75,121-134)
    def vertices(self): return self._V.copy()
    def nodes(self): return self._h.node_positions()
    def elements(self): return self._F.copy()
    def elementNodes(self): return self._h.elem_nodes()
    def boundaryElements(self): return self._h.boundary_elem_nodes()[:, :self._K]
    def boundaryNodes(self): return self._h.boundary_nodes()
    def numVertices(self): return len(self._V)
    def numElements(self): return len(self._F)
    def numNodes(self): return self._h.n_node
    def copy(self): return _Mesh(self._V, self._F, self._deg, self._N)

    def setVertices(self, V):
        V = np.ascontiguousarray(np.asarray(V, dtype=np.float64)[:, :self._N])
        self._h.mesh_update_vertices(V)                      # same connectivity: no topology rebuild
        self._V = V

    def elementVolumes(self):
        P = self._V[self._F]
        if self._K == 3:
            return np.linalg.det(P[:, 1:] - P[:, :1]) / 6.0
        e1, e2 = P[:, 1] - P[:, 0], P[:, 2] - P[:, 0]
        return 0.5 * (e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0])

    def barycenters(self): return self._V[self._F].mean(axis=1)

    def save(self, path):
        mesh_io.MSHFieldWriter(path, self._V, self._F).close()

    def field_writer(self, path):
        return mesh_io.MSHFieldWriter(path, self._V, self._F)

    @staticmethod
    def is_tet_mesh(): return None

    @property
    def bbox(self): return self._V.min(axis=0), self._V.max(axis=0)
    @property
    def bbox_volume(self): return float(np.prod(self._V.max(axis=0) - self._V.min(axis=0)))
    @property
    def volume(self): return float(self.elementVolumes().sum())
    @property
    def degree(self): return self._deg
    @property
    def simplexDimension(self): return self._K
    @property
    def embeddingDimension(self): return self._N


def Mesh(*args, degree=1, embeddingDimension=3, **kw):
    """mesh.cc:293-330: `Mesh(path, degree=1, embeddingDimension=3)` or `Mesh(V, F, degree=1, embeddingDimension=3)`."""
    args = list(args)
    if "path" in kw:
        args.insert(0, kw.pop("path"))
    if "V" in kw:
        args = [kw.pop("V"), kw.pop("F")] + args
    if isinstance(args[0], str):
        V, F, _ = mesh_io.load_msh(args[0])
        rest = args[1:]
    else:
        V, F = args[0], args[1]
        rest = args[2:]
    if len(rest) > 0: degree = rest[0]
    if len(rest) > 1: embeddingDimension = rest[1]
    if kw:
        raise TypeError("unexpected arguments: %s" % sorted(kw))
    return _Mesh(V, F, degree, embeddingDimension)


class _PeriodicCondition:
    def __init__(self, m, eps, ignore_mismatch=False, ignore_dims=()):
        h = Context(-1)
        h.mesh_build(m._F, m._V, m._deg)
        h.set_option("periodic_ignore_mismatch", 1 if ignore_mismatch else 0)
        h.set_option("periodic_ignore_dims", sum(1 << int(d) for d in ignore_dims))
        self._n = h.apply_periodic_conditions(eps)
        self._dofs = h.get_dof_map()[0]
        h.close()

    def periodicDoFsForNodes(self): return self._dofs.copy()
    def numPeriodicDoFs(self): return self._n


def PeriodicCondition(mesh, eps=1e-7, ignore_mismatch=False, ignore_dims=()):
    return _PeriodicCondition(mesh, eps, ignore_mismatch, ignore_dims)
