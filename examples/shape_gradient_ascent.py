"""Shape-gradient loop on the GPU path: raise one entry of the homogenized elasticity tensor of a periodic cell by moving
its interior vertices along the exact discrete gradient.

    python examples/shape_gradient_ascent.py [mesh.msh] [steps]

Every step uses the pieces the reference's shape optimisers use (PeriodicHomogenization.hh:372-480, LinearElasticity.hh:1279):
  solve the 3 / 6 cell problems            (PCG, matrix and preconditioner on the device)
  Ch  = energy form of the homogenized tensor               (mfh_mutual_energies)
  dCh = its derivative w.r.t. every vertex coordinate        (mfh_mutual_energy_differential, one element sweep per entry)
  move the interior vertices, keep the connectivity          (mfh_mesh_update_vertices: no topology / symbolic rebuild)
The boundary of the cell is kept fixed so that the cell stays periodic."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshfem_amd import homogenization as H, mesh_io          # noqa: E402
from meshfem_amd.linear_elasticity import Simulator            # noqa: E402


def run(mesh_path, steps=3, entry=(0, 0), step_size=0.01, degree=2, verbose=True):
    V, E, _ = mesh_io.load_mesh(mesh_path)
    N = E.shape[1] - 1
    V = np.ascontiguousarray(V[:, :N])
    fl = N * (N + 1) // 2
    pair = sum(fl - k for k in range(entry[0])) + (entry[1] - entry[0])      # row-major index over the upper triangle
    sim = Simulator(E, V, degree)
    sim.rtol = 1e-11
    sim.setIsotropicMaterial(200.0, 0.35)
    interior = ~((np.abs(V - V.min(axis=0)) < 1e-9).any(axis=1) | (np.abs(V - V.max(axis=0)) < 1e-9).any(axis=1))
    h = np.sqrt(np.prod(V.max(axis=0) - V.min(axis=0)) / len(E))            # a typical element size
    history = []
    for it in range(steps + 1):
        w, _ = H.solve_cell_problems(sim)
        Ch = H.homogenized_elasticity_tensor_energy_form(sim, w)
        g = H.homogenized_elasticity_tensor_discrete_differential(sim, w)[pair] * interior[:, None]
        history.append(dict(value=float(Ch[entry]), gradient_norm=float(np.linalg.norm(g)), symbolic_ms=sim.ctx.timing()["symbolic_ms"]))
        if verbose:
            print("step %d: Ch[%d,%d] = %.8f   |grad| = %.4e" % (it, entry[0], entry[1], Ch[entry], history[-1]["gradient_norm"]))
        if it == steps:
            break
        V = V + step_size * h * g / np.abs(g).max()
        sim.updateMeshNodePositions(V)
    return history, sim, V, g


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    mesh = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "tests", "golden", "meshes", "2D_microstructure.msh")
    run(mesh, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
