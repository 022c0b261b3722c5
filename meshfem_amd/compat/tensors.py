"""`tensors` (src/python_bindings/tensors.cc:19-75,108-109): ElasticityTensor2D/3D and SymmetricMatrix."""
import numpy as np

from meshfem_amd.tensors import ElasticityTensor as _Base, flat_len


class _ElasticityTensor(_Base):
    def setIdentity(self):                                    # ElasticityTensor.hh: 4th-order symmetric identity
        n, d = flat_len(self.dim), self.dim
        self.D = np.zeros((n, n))
        self.D[np.arange(d), np.arange(d)] = 1.0
        self.D[np.arange(d, n), np.arange(d, n)] = 0.5
        return self

    def _idx(self, i, j):
        d = self.dim
        return i if i == j else (d * (d + 1) // 2 - i - j)      # Flattening.hh:47-60

    def __call__(self, i, j, k, l):                           # tensors.cc:30-34
        if max(i, j, k, l) >= self.dim:
            raise RuntimeError("Index out of bounds")
        return float(self.D[self._idx(i, j), self._idx(k, l)])

    def inverse(self):
        """Compliance tensor in the same (tensor-shear) flattening: S = W^-1 D^-1 W^-1 with W the
        shear-doubling of the double contraction (ElasticityTensor.hh:437-449)."""
        d, n = self.dim, flat_len(self.dim)
        w = np.ones(n); w[d:] = 2.0
        out = type(self)(self.dim)
        out.D = np.linalg.inv(self.D * w[None, :]) / w[None, :]
        return out

    def frobeniusNormSq(self):
        d, n = self.dim, flat_len(self.dim)
        w = np.ones(n); w[d:] = 2.0
        return float(np.sum(self.D ** 2 * np.outer(w, w)))

    def __sub__(self, other):
        out = type(self)(self.dim)
        out.D = self.D - other.D
        return out

    def __repr__(self):
        return "ElasticityTensor%dD(\n%s)" % (self.dim, np.array2string(self.D, precision=6))


class ElasticityTensor3D(_ElasticityTensor):
    def __init__(self, E=1.0, nu=0.0):                       # tensors.cc:21-23 default (E=1, nu=0)
        super().__init__(3, E, nu)


class ElasticityTensor2D(_ElasticityTensor):
    def __init__(self, E=1.0, nu=0.0):
        super().__init__(2, E, nu)


class _SymmetricMatrixValue:
    def __init__(self, dim, flat):
        self.N, self.flat = dim, np.asarray(flat, dtype=np.float64)

    def __call__(self, i, j):
        if i >= self.N or j >= self.N:
            raise RuntimeError("Index out of bounds")
        d = self.N
        return float(self.flat[i if i == j else (d * (d + 1) // 2 - i - j)])

    def __getitem__(self, k):
        return float(self.flat[k])

    def toMatrix(self):
        return np.array([[self(i, j) for j in range(self.N)] for i in range(self.N)])

    def eigenvalues(self):
        return np.linalg.eigvalsh(self.toMatrix())


def SymmetricMatrix(values):
    """tensors.cc:108-109: from the flattened values (3 or 6) or from an N x N matrix."""
    a = np.asarray(values, dtype=np.float64)
    if a.ndim == 2:
        d = a.shape[0]
        flat = np.zeros(flat_len(d))
        for i in range(d):
            for j in range(i, d):
                flat[i if i == j else (d * (d + 1) // 2 - i - j)] = a[i, j]
        return _SymmetricMatrixValue(d, flat)
    return _SymmetricMatrixValue({3: 2, 6: 3}[a.size], a)
