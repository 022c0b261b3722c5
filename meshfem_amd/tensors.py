"""Host-side ElasticityTensor helper mirroring src/python_bindings/tensors.cc:19-75
(`ElasticityTensor3D(E, nu)`, `.setOrthotropic(...)`, `.D`): builds the flattened tensor that
`Context.material_const` / `material_tensor_field` take. Voigt order xx,yy,zz,yz,xz,xy
(Flattening.hh:47-60), TENSOR shear entries (ElasticityTensor.hh:100-164)."""
import numpy as np


def flat_len(dim):
    return dim * (dim + 1) // 2


class ElasticityTensor:
    def __init__(self, dim, E=1.0, nu=0.3):
        self.dim = dim
        self.D = np.zeros((flat_len(dim),) * 2)
        self.setIsotropic(E, nu)

    def setIsotropic(self, E, nu):                       # ElasticityTensor.hh:100-134 (2D: plane stress)
        lam = (nu * E) / ((1.0 + nu) * (1.0 - 2.0 * nu))
        if self.dim == 2:
            lam = (nu * E) / (1.0 - nu * nu)
        mu = E / (2.0 + 2.0 * nu)
        d, n = self.dim, flat_len(self.dim)
        self.D = np.zeros((n, n))
        self.D[:d, :d] = lam
        self.D[np.arange(d), np.arange(d)] = lam + 2 * mu
        self.D[np.arange(d, n), np.arange(d, n)] = mu
        return self

    def setOrthotropic(self, *p):                        # ElasticityTensor.hh:136-164
        if self.dim == 3:
            Ex, Ey, Ez, nuYX, nuZX, nuZY, muYZ, muZX, muXY = p
            m = np.zeros((6, 6))
            m[0, 0], m[0, 1], m[0, 2] = 1.0 / Ex, -nuYX / Ey, -nuZX / Ez
            m[1, 1], m[1, 2], m[2, 2] = 1.0 / Ey, -nuZY / Ez, 1.0 / Ez
            m[3, 3], m[4, 4], m[5, 5] = 1.0 / muYZ, 1.0 / muZX, 1.0 / muXY
        else:
            Ex, Ey, nuYX, muXY = p
            m = np.zeros((3, 3))
            m[0, 0], m[0, 1], m[1, 1], m[2, 2] = 1.0 / Ex, -nuYX / Ey, 1.0 / Ey, 1.0 / muXY
        m = np.triu(m) + np.triu(m, 1).T
        self.D = np.linalg.inv(m)
        return self

    def setIsotropicLame(self, lam, mu):                 # ElasticityTensor.hh:117-134
        d, n = self.dim, flat_len(self.dim)
        self.D = np.zeros((n, n))
        self.D[:d, :d] = lam
        self.D[np.arange(d), np.arange(d)] = lam + 2 * mu
        self.D[np.arange(d, n), np.arange(d, n)] = mu
        return self

    def _doubler(self):
        s = np.ones(flat_len(self.dim))
        s[self.dim:] = 2.0
        return s

    def __call__(self, i, j, k, l):                      # ElasticityTensor.hh:274-277
        return self.D[_flat(self.dim, i, j), _flat(self.dim, k, l)]

    def inverse(self):                                   # ElasticityTensor.hh:315-323: S^-1 F(E)^-1 S^-1
        out = ElasticityTensor(self.dim)
        s = self._doubler()
        out.D = np.linalg.inv(self.D) / s[:, None] / s[None, :]
        return out

    def computeEigenstrains(self):                       # ElasticityTensor.hh:555-579: (lambdas ascending, strains as columns)
        r = np.sqrt(self._doubler())
        lam, Q = np.linalg.eigh(r[:, None] * self.D * r[None, :])
        return lam, Q / r[:, None]

    def getOrthotropicParameters(self):                  # ElasticityTensor.hh:166-229
        S = self.inverse().D
        if self.dim == 2:                                # Ex Ey nuYX muXY
            return [1.0 / S[0, 0], 1.0 / S[1, 1], -S[0, 1] / S[1, 1], 0.25 / S[2, 2]]
        Ex, Ey, Ez = 1.0 / S[0, 0], 1.0 / S[1, 1], 1.0 / S[2, 2]
        return [Ex, Ey, Ez, -S[0, 1] * Ey, -S[0, 2] * Ez, -S[1, 2] * Ez, 0.25 / S[3, 3], 0.25 / S[4, 4], 0.25 / S[5, 5]]

    def anisotropy(self):                                # ElasticityTensor.hh:251-268
        p = self.getOrthotropicParameters()
        if self.dim == 2:
            E_avg, nu_avg, mu_avg = (p[0] + p[1]) / 2.0, p[2], p[3]
        else:
            E_avg, nu_avg, mu_avg = sum(p[0:3]) / 3.0, sum(p[3:6]) / 3.0, sum(p[6:9]) / 3.0
        return mu_avg / (E_avg / (2 * (1 + nu_avg)))

    def quadrupleContract(self, other):                  # ElasticityTensor.hh:498-506
        s = self._doubler()
        return float(np.sum(self.D * other.D * s[:, None] * s[None, :]))

    def frobeniusNormSq(self):                           # :508
        return self.quadrupleContract(self)

    def __sub__(self, other):
        out = ElasticityTensor(self.dim)
        out.D = self.D - other.D
        return out

    def writeUnflattened(self):                          # ElasticityTensor.hh:613-633 (Mathematica array syntax)
        d = self.dim
        fmt = lambda x: repr(float(x))
        return "{" + ", ".join("{" + ", ".join("{" + ", ".join("{" + ", ".join(fmt(self(i, j, k, l)) for l in range(d)) + "}"
                                                                for k in range(d)) + "}" for j in range(d)) + "}" for i in range(d)) + "}"

    def doubleContract(self, flat_strain):               # ElasticityTensor.hh:437-449
        e = np.array(flat_strain, dtype=np.float64)
        e[self.dim:] *= 2.0
        return self.D @ e


def _flat(dim, i, j):                                    # Flattening.hh:47-60
    if i == j:
        return i
    return flat_len(dim) - i - j


def closest_isotropic_tensor(C):
    """closestIsotropicTensor (TensorProjection.hh:22-75): projection onto span{J, K} in the Frobenius metric."""
    n = C.dim
    C_ijij = sum(C(i, j, i, j) for i in range(n) for j in range(n))
    C_iijj = sum(C(i, i, j, j) for i in range(n) for j in range(n))
    CdotJ = C_iijj / n
    CdotK = C_ijij - CdotJ
    alpha, beta = CdotJ, CdotK / (0.5 * (n * n + n) - 1.0)
    return ElasticityTensor(n).setIsotropicLame((alpha - beta) / n, beta / 2.0)


def ElasticityTensor3D(E=1.0, nu=0.3):
    return ElasticityTensor(3, E, nu)


def ElasticityTensor2D(E=1.0, nu=0.3):
    return ElasticityTensor(2, E, nu)
