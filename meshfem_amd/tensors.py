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

    def doubleContract(self, flat_strain):               # ElasticityTensor.hh:437-449
        e = np.array(flat_strain, dtype=np.float64)
        e[self.dim:] *= 2.0
        return self.D @ e


def ElasticityTensor3D(E=1.0, nu=0.3):
    return ElasticityTensor(3, E, nu)


def ElasticityTensor2D(E=1.0, nu=0.3):
    return ElasticityTensor(2, E, nu)
