"""`LinearElasticity.Simulator` with the reference's method names
(src/lib/MeshFEM/LinearElasticity.hh:434-1659) on top of the C ABI. Every numeric step runs in
libmeshfem_hip.so on the GPU; this class only forwards."""
import numpy as np

from . import _lib as L
from .core import Context, flat_len


class Simulator:
    def __init__(self, elems, vertices, degree=2, device=0):
        """== Simulator(elems, vertices) (:460-473): builds the FEMMesh, raises on inverted elements."""
        self.ctx = Context(device)
        self.ctx.mesh_build(elems, vertices, degree)
        self.N = self.ctx.dim
        self.degree = degree
        self.rtol, self.maxit = 1e-8, 100000
        self._use_pin = False
        self._no_rigid_motion = False
        self._rigid_motion_rhs = None
        self._bbox = None

    # ---- mesh queries (mesh.cc:47-70)
    def numNodes(self):
        return self.ctx.n_node

    def numElements(self):
        return self.ctx.n_elem

    def numDoFs(self):
        return self.ctx.n_dof

    def nodes(self):
        return self.ctx.node_positions()

    def elements(self):
        return self.ctx.elem_nodes()

    def boundingBox(self):
        """(min corner, max corner) of the mesh (BoundaryConditions.hh:452-470 builds the periodic cell from it); edge nodes are
        midpoints, so the box of the nodes is the box of anything that contains the vertices. Column by column: numpy's axis-0
        reduction of an (n, 3) array is three times slower."""
        if self._bbox is None:
            pos = self.nodes()
            self._bbox = (np.array([pos[:, k].min() for k in range(pos.shape[1])]), np.array([pos[:, k].max() for k in range(pos.shape[1])]))
        return self._bbox

    def updateMeshNodePositions(self, vertices):                        # :1279-1284
        self.ctx.mesh_update_vertices(vertices)
        self._bbox = None

    # ---- materials
    def setMaterial(self, tensor):
        self.ctx.material_const(np.asarray(tensor.D if hasattr(tensor, "D") else tensor))

    def setIsotropicMaterial(self, E, nu):
        self.ctx.material_isotropic(E, nu)

    def setIsotropicField(self, E, nu):
        self.ctx.material_iso_field(E, nu)

    def setOrthotropicField(self, params):
        self.ctx.material_ortho_field(params)

    # ---- boundary conditions (box regions of applyBoundaryConditions, :881-1027)
    def applyDirichletBox(self, mn, mx, value, relative=False, components=None):
        self.ctx.bc_dirichlet_box(mn, mx, value, relative, components)

    def applyNeumannBox(self, mn, mx, value, kind=L.NEUMANN_TRACTION, relative=False):
        self.ctx.bc_neumann_box(mn, mx, value, kind, relative)

    def applyDirichletNodes(self, nodes, values, components=None):      # DirichletNodesCondition, :991-1002
        self.ctx.bc_dirichlet_nodes(nodes, values, components)

    def applyNeumannElements(self, bdry_elems, tractions):             # NeumannElementsCondition, :966-990
        self.ctx.bc_neumann_elements(bdry_elems, tractions)

    def applyPeriodicConditions(self, epsilon=1e-7, ignoreMismatch=False, ignoreDims=()):    # :845-854
        self.ctx.set_option("periodic_ignore_mismatch", 1 if ignoreMismatch else 0)
        self.ctx.set_option("periodic_ignore_dims", sum(1 << int(d) for d in ignoreDims))
        return self.ctx.apply_periodic_conditions(epsilon)

    def removePeriodicConditions(self):                                 # :874-879
        self.ctx.dof_map(None, 0)

    def applyNoRigidMotionConstraint(self):                             # m_useRigidMotionConstraint (:1214-1228)
        self._no_rigid_motion = True

    def removeNoRigidMotionConstraint(self):
        self._no_rigid_motion = False

    def setRigidMotionConstraintRHS(self, rhs):
        self._rigid_motion_rhs = None if rhs is None else np.asarray(rhs, dtype=np.float64)

    def setUsePinNoRigidTranslationConstraint(self, use):               # PeriodicHomogenization.hh:44-45
        self._use_pin = bool(use)

    # ---- loads / solve
    def neumannLoad(self):                                              # :703-717
        return self.ctx.neumann_load()

    def constantStrainLoad(self, cstrain_flat):                         # :551-562
        return self.ctx.constant_strain_load(cstrain_flat)

    def solve(self, f=None):                                            # :479-487, :657
        f = None if f is None else np.asarray(f, dtype=np.float64).ravel()
        flags = (L.SOLVE_PIN if self._use_pin else 0) | (L.SOLVE_NO_RIGID_MOTION if self._no_rigid_motion else 0)
        u = self.ctx.sim_solve_constrained(f, flags, self._rigid_motion_rhs, rtol=self.rtol, maxit=self.maxit)
        self.info = dict(self.ctx.last_info)
        return u

    def solveMany(self, loads):
        """solve() for several load vectors on the same constrained system (what solveCellProblems does with its constantStrainLoad vectors,
        PeriodicHomogenization.hh:47-53): returns the list of nodal displacement fields; self.infos holds one record per load."""
        F = np.stack([np.asarray(f, dtype=np.float64).ravel() for f in loads])
        flags = (L.SOLVE_PIN if self._use_pin else 0) | (L.SOLVE_NO_RIGID_MOTION if self._no_rigid_motion else 0)
        if self._rigid_motion_rhs is not None:          # (a right-hand side for the constraint rows: one load after the other)
            out, self.infos = [], []
            for f in F:
                out.append(self.solve(f))
                self.infos.append(self.info)
            return out
        u, infos = self.ctx.sim_solve_batch(F, flags, rtol=self.rtol, maxit=self.maxit)
        self.infos = infos
        self.info = dict(infos[-1])
        return [u[k] for k in range(len(F))]

    def solveConstantStrainLoads(self, cstrains):
        """[solve(constantStrainLoad(e)) for e in cstrains] in one call (the loop of solveCellProblems, PeriodicHomogenization.hh:47-53): the load
        vectors never visit the host where the library can form them on the device; self.infos holds one record per strain."""
        if self._rigid_motion_rhs is not None:
            return self.solveMany([self.constantStrainLoad(e) for e in cstrains])
        flags = (L.SOLVE_PIN if self._use_pin else 0) | (L.SOLVE_NO_RIGID_MOTION if self._no_rigid_motion else 0)
        w, infos = self.ctx.solve_cell_problems(np.asarray(cstrains, dtype=np.float64), flags, rtol=self.rtol, maxit=self.maxit)
        self.infos = infos
        self.info = dict(infos[-1])
        return [w[k] for k in range(len(w))]

    def applyStiffnessMatrix(self, u_dofs):                             # :801-823
        return self.ctx.apply_K(np.asarray(u_dofs).ravel()).reshape(-1, self.N)

    def strainField(self, u_nodes):                                     # :511-517 (interpolant values per element)
        return self.ctx.strain_field(u_nodes)

    def stressField(self, u_nodes):                                     # :519-526
        return self.ctx.strain_field(u_nodes, stress=True)

    def elementStrain(self, i, u_nodes):                                # :493-497 (one element of strainField)
        return self.ctx.strain_field(u_nodes)[i]

    def averageStrainField(self, u_nodes):                              # :528-538
        return self.ctx.average_strain(u_nodes)

    def boundaryStrainField(self, u_nodes, stress=False):
        """strain (stress) interpolants restricted to the boundary elements (restrictInterpolant, InterpolantRestriction.hh:29-66):
        [nBdryElem, 1 | N, flatLen], values at the boundary element's corners in its own vertex order"""
        return self.ctx.boundary_strain_field(u_nodes, stress=stress)

    def averageStressField(self, u_nodes):                              # :539-549
        return self.ctx.average_stress(u_nodes)

    # ---- discrete shape derivatives (forward mode), :1297-1374
    def applyDeltaStiffnessMatrix(self, u_nodes, deltaP):               # :1301-1328  per-node u -> per-DoF load
        return self.ctx.apply_delta_K(u_nodes, deltaP)

    def deltaConstantStrainLoad(self, cstrain_flat, deltaP):            # :1331-1348
        return self.ctx.delta_constant_strain_load(cstrain_flat, deltaP)

    def deltaAverageStrainField(self, u_nodes, deltaU, deltaP):         # :1364-1374
        return self.ctx.delta_average_strain(u_nodes, deltaU, deltaP)

    def deltaAverageStressField(self, u_nodes, deltaU, deltaP):         # C : deltaAverageStrainField (deltaStress :280-286)
        return self.ctx.delta_average_strain(u_nodes, deltaU, deltaP, stress=True)

    def benchmarkReport(self):
        """Timings under the reference's timer-section names (GlobalBenchmark.hh:14-34; sections of
        LinearElasticity.hh:1206,1394-1399,482-485 and SparseMatrices.hh:283): milliseconds of the last operations.
        "Compress Matrix" is the once-per-mesh symbolic phase here (sumRepeated's sort/merge hoisted out of the
        numeric assembly); the CHOLMOD sections have no counterpart (PCG)."""
        t = self.ctx.timing()
        info = getattr(self, "info", None) or {}
        return {"Assemble System": t["geometry_ms"] + t["assemble_ms"], "Compress Matrix": t["symbolic_ms"],
                "Set System": t["upload_ms"], "Fix Variables": info.get("setup_ms", 0.0),
                "Elasticity Solve": info.get("solve_ms", 0.0), "PCG iterations": info.get("iterations", 0)}

    def assembleStiffnessMatrix(self):
        """m_assembleStiffnessMatrix (:1408-1466) + sumRepeated: upper triplets (i, j, v)."""
        self.ctx.assemble()
        return self.ctx.export_upper_triplets()

    def stiffnessMatrix(self):
        self.ctx.assemble()
        return self.ctx.export_scipy()
