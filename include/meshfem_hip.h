/*
 * meshfem_hip.h -- C ABI of libmeshfem_hip.so: the MI355X (gfx950) implementation of MeshFEM's
 * per-element linear-elasticity stiffness assembly + sparse solve hot path.
 *
 * The reference (MeshFEM) has no FFI seam for this path: everything is C++ templates
 * instantiated in the caller.  Every entry point below therefore cites the
 * reference member function it replaces (paths relative to src/lib/MeshFEM/).  The C++ facade
 * (include/MeshFEMHip/LinearElasticity.hh) and the Python mirror (meshfem_amd/) are thin
 * wrappers around exactly these symbols.
 *
 * Conventions
 *   - plain pointers and sizes only; caller owns every host buffer, the library owns device memory
 *   - every function returns mfh_status (0 = ok); mfh_last_error(ctx) gives the message that the
 *     C++ facade rethrows as std::runtime_error (the reference's error convention)
 *   - Real = double everywhere (Types.hh:8); vector fields are interleaved [x0 y0 z0 x1 ...]
 *     (Fields.hh:15-17,49); local node order is MeshFEM/Gmsh order (Simplex.hh:30-47)
 *   - one context per host thread / device; no hidden global state (the reference's static
 *     HomogenousMaterialGetter material, LinearElasticity.hh:31-39,426-427, is per-context here)
 *   - "_dev" entry points take DEVICE pointers (HIP allocations, e.g. torch tensors) and enqueue on
 *     the context's stream; they exist for the multi-GPU driver that runs RCCL collectives between
 *     the local kernels
 */
#ifndef MESHFEM_HIP_H
#define MESHFEM_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mfh_ctx mfh_ctx;
typedef int32_t mfh_status;

enum {
    MFH_OK = 0,
    MFH_ERR_INVALID = 1,   /* bad argument / bad mesh (e.g. negatively oriented element) */
    MFH_ERR_STATE = 2,     /* call order violated (e.g. solve before mesh) */
    MFH_ERR_HIP = 3,       /* HIP runtime failure (no device, OOM, launch failure) */
    MFH_ERR_NOT_CONVERGED = 4,
    MFH_ERR_UNSUPPORTED = 5
};

/* assembly strategies (north_star: "colored or atomic" scatter, chosen by rocprof evidence) */
enum {
    MFH_ASSEMBLE_GATHER = 0, /* owner-computes row chunks, LDS accumulation, K written once    */
    MFH_ASSEMBLE_ATOMIC = 1  /* element-major, global_atomic_add_f64 scatter (baseline variant) */
};

/* Neumann condition kinds (BoundaryConditions.hh NeumannType; LinearElasticity.hh:908-931) */
enum { MFH_NEUMANN_TRACTION = 0, MFH_NEUMANN_PRESSURE = 1, MFH_NEUMANN_FORCE = 2 };

/* operator assembled into K. The scalar operators run on the same machinery with 1x1 blocks: one variable per
 * DoF, vectors of length nDoF, `mfh_element_stiffness` gives npe x npe matrices, Dirichlet boxes use value[0]
 * (PoissonMesh::applyBoundaryConditions, Poisson.hh:57-89), `mfh_sim_solve(f = NULL)` solves with a zero
 * right-hand side (Poisson.hh:91-117).
 *   MFH_OP_LAPLACIAN  int grad phi_i . grad phi_j   (Laplacian::construct, Laplacian.hh:27-57; full degree)
 *   MFH_OP_MASS       int phi_i phi_j               (MassMatrix::construct, MassMatrix.hh:50-86; full degree, not lumped) */
enum { MFH_OP_ELASTICITY = 0, MFH_OP_LAPLACIAN = 1, MFH_OP_MASS = 2 };

/* preconditioners */
enum {
    MFH_PRECOND_BLOCK_JACOBI = 0, MFH_PRECOND_JACOBI = 1, MFH_PRECOND_NONE = 2,
    /* block-Jacobi + additive coarse correction on per-aggregate rigid-body modes (falls back to
     * block-Jacobi for partitioned rows and the scalar operators; see mfh_precond_info) */
    MFH_PRECOND_TWO_LEVEL = 3,
    /* p-multigrid V-cycle for quadratic elasticity: Chebyshev-smoothed quadratic level (matrix-free operator) -> linear level on the
     * same vertices (its assembled stiffness matrix IS the Galerkin operator: P1 is a subspace of P2) -> rigid-body modes of
     * aggregates, merged 2^dim at a time down to a dense inverse. Linear elements enter the same hierarchy at its linear level (their own
     * assembled K). Mesh-independent iteration counts (tens instead of hundreds); several right-hand sides share the coarse levels of every
     * V-cycle (option "mg_batch"). On a row-partitioned context (mfh_dist_solve, the preconditioner set on EVERY rank) the nodal levels are
     * partitioned like the mesh and the aggregate levels partitioned or replicated by size; scalar operators fall back to
     * MFH_PRECOND_TWO_LEVEL (see mfh_precond_info). Options "mg_steps_fine", "mg_steps_coarse", "mg_ratio_fine", "mg_ratio_coarse", "mg_agg_nodes". */
    MFH_PRECOND_MULTIGRID = 4,
    /* the drivers' default: MFH_PRECOND_MULTIGRID unless the mesh as a whole is stretched by more than option "auto_stretch_max" (default 8:
     * sqrt of the ratio of the extreme eigenvalues of the mesh's edge covariance sum_e sum_edges e e^T, reduced on the device -- 1 for an isotropic
     * mesh whatever the shapes of its elements, s for a mesh stretched s : 1 : 1), where MFH_PRECOND_TWO_LEVEL is the faster one (measured
     * crossover, docs/design/04_4c_multigrid.md). The reference's direct solve knows no such cliff (SparseMatrices.hh:1984-2296); the choice is
     * made when a solve prepares itself and again after new vertices; mfh_precond_choice reads it. */
    MFH_PRECOND_AUTO = 5
};

typedef struct mfh_solve_info {
    int32_t iterations;      /* PCG iterations of the last right-hand side          */
    int32_t converged;       /* 1 if ||r||/||b|| <= rtol                            */
    double  rel_residual;    /* recurrence residual ||r||_2 / ||b||_2 at exit       */
    double  true_rel_residual; /* ||b - A x||_2 / ||b||_2 recomputed at exit        */
    double  solve_ms;        /* device time of the PCG loop (HIP events), last RHS  */
    double  setup_ms;        /* device time of rhs/preconditioner setup             */
    int32_t used_graph;      /* 1 if the PCG iterations were replayed from a captured hipGraph */
    int32_t reserved;
} mfh_solve_info;

typedef struct mfh_timing {
    double symbolic_ms;      /* host: BSR pattern + gather/scatter maps (once per mesh/DoF map) */
    double geometry_ms;      /* device: element embedding kernel (a1)                           */
    double assemble_ms;      /* device: last numeric assembly launch (a4+a6+a9)                 */
    double upload_ms;        /* host->device copies of the last mesh_set/build                  */
} mfh_timing;

/* ---------------------------------------------------------------- context
 * device >= 0: HIP device ordinal (fails with MFH_ERR_HIP when there is none: NO CPU fallback).
 * device == -1: host-only context; only the mesh / DoF-map / symbolic / boundary-condition host
 * logic works, every numeric entry point returns MFH_ERR_HIP. Used by the CPU test-suite.        */
mfh_status  mfh_create(int32_t device, mfh_ctx** out);
void        mfh_destroy(mfh_ctx* ctx);
const char* mfh_last_error(const mfh_ctx* ctx);
const char* mfh_version(void);
/* the HIP stream all work of this context is enqueued on (a hipStream_t) */
void*       mfh_stream(mfh_ctx* ctx);
/* adopt a caller-owned hipStream_t (e.g. torch's current stream) so that the _dev entry points are
 * stream-ordered with the caller's own kernels and RCCL collectives; the context never destroys it */
mfh_status  mfh_set_stream(mfh_ctx* ctx, void* hipStream);

/* ---------------------------------------------------------------- mesh
 * mfh_mesh_build   == FEMMesh(elems, vertices) ctor (FEMMesh.inl:11-82): vertex nodes = vertices,
 *                     edge nodes numbered nVert+k in first-encounter order over (element, local
 *                     edge) (FEMMesh.inl:22-36), edge node = midpoint (FEMMesh.hh:228-233);
 *                     boundary elements/vertices/nodes in the reference's enumeration order
 *                     (TetMesh.inl:36-91, TriMesh.inl:86-118, FEMMesh.inl:39-59); embeds every
 *                     element (FEMMesh.hh:58-66 -> EmbeddedElement.hh:162-241) and fails with
 *                     MFH_ERR_INVALID on a negative volume (LinearElasticity.hh:465-472).
 * mfh_mesh_set     == the same with the node table supplied by the caller (no topology built:
 *                     boundary-condition helpers are unavailable). nOwned < nNode restricts K to
 *                     the first nOwned block rows (row-partitioned multi-GPU; halo nodes last).
 * dim in {2,3}; deg in {1,2}; simplices only (tri if dim==2, tet if dim==3).                  */
mfh_status mfh_mesh_build(mfh_ctx* ctx, int32_t dim, int32_t deg, int64_t nElem, int64_t nVert,
                          const int32_t* elemVerts /* nElem x (dim+1) */,
                          const double* vertPos /* nVert x dim */);
mfh_status mfh_mesh_set(mfh_ctx* ctx, int32_t dim, int32_t deg, int64_t nElem, int64_t nNode, int64_t nOwned,
                        const int32_t* elemNodes /* nElem x nodesPerElem, MeshFEM local order, any numbering */,
                        const double* nodePos /* nNode x dim (only corner-node positions enter the embedding) */);
mfh_status mfh_mesh_sizes(const mfh_ctx* ctx, int64_t* nElem, int64_t* nNode, int64_t* nVert,
                          int64_t* nBdryElem, int64_t* nBdryNode, int32_t* nodesPerElem,
                          int32_t* nodesPerBdryElem);
mfh_status mfh_mesh_get_elem_nodes(const mfh_ctx* ctx, int32_t* out /* nElem x nodesPerElem */);
/* == Simulator::updateMeshNodePositions (LinearElasticity.hh:1279-1284; FEMMesh::setNodePositions): new vertex positions
 * [nVert x dim] on the SAME connectivity. Elements are re-embedded at the next assembly / solve; topology, boundary
 * numbering, DoF map, sparsity pattern, gather lists and matrix-free lists are kept (they depend on connectivity only), so a
 * shape-optimisation step costs one vertex upload + geometry kernel + numeric assembly instead of a mesh rebuild.
 * Boundary areas / normals are recomputed; boundary-condition VALUES already stored (tractions of pressure regions) are
 * not, exactly like the reference. Contexts built by mfh_mesh_set (row-partitioned ones among them) pass the positions of ALL
 * their nodes [nNode x dim], halo nodes with the coordinates their owners hold; the exchange lists of mfh_dist_setup stay. */
mfh_status mfh_mesh_update_vertices(mfh_ctx* ctx, const double* vertPos);
mfh_status mfh_mesh_get_node_positions(const mfh_ctx* ctx, double* out /* nNode x dim */);
mfh_status mfh_mesh_get_boundary_elem_nodes(const mfh_ctx* ctx, int32_t* out /* nBE x npbe, volume node ids */);
mfh_status mfh_mesh_get_boundary_nodes(const mfh_ctx* ctx, int32_t* out /* nBdryNode volume node ids */);
/* the volume element each boundary element is a face / edge of (BoundaryElementHandle -> opposite simplex) */
mfh_status mfh_mesh_get_boundary_elem_parents(const mfh_ctx* ctx, int32_t* out /* nBE */);
/* BoundaryElementData::isInternal: 1 on the periodic cell boundary once a periodic condition is installed (else 0) */
mfh_status mfh_mesh_get_boundary_elem_internal(const mfh_ctx* ctx, uint8_t* out /* nBE */);
mfh_status mfh_mesh_get_boundary_elem_geometry(const mfh_ctx* ctx, double* volume /* nBE */,
                                               double* normal /* nBE x dim */);
mfh_status mfh_mesh_get_elem_volumes(mfh_ctx* ctx, double* out /* nElem */);

/* ---------------------------------------------------------------- materials
 * mfh_material_const       == HomogenousMaterialGetter / Materials::Constant::setTensor
 *                             (LinearElasticity.hh:31-39): one flattened D (flatLen x flatLen,
 *                             Voigt xx,yy,zz,yz,xz,xy, Flattening.hh:47-60; tensor shear, not
 *                             engineering) for every element.
 * mfh_material_isotropic   == ElasticityTensor::setIsotropic(E,nu) (ElasticityTensor.hh:100-134;
 *                             2D is plane stress), constant.
 * mfh_material_iso_field   == per-element E,nu (ETensorStoreGetter, LinearElasticity.hh:20-29;
 *                             Simulate_cli.cc:116-135).
 * mfh_material_ortho_field == per-element setOrthotropic3D/2D (ElasticityTensor.hh:136-164),
 *                             params per element: 3D Ex,Ey,Ez,nuYX,nuZX,nuZY,muYZ,muZX,muXY;
 *                             2D Ex,Ey,nuYX,muXY. The compliance inverse runs on the device; the element
 *                             record keeps only the normal block and the shear stiffnesses of D (192 B).
 * mfh_material_tensor_field== per-element flattened D (flatLen x flatLen each).
 * Per-element isotropic / orthotropic parameters that give an indefinite tensor (E <= 0, nu outside (-1, 1/2), a
 * compliance matrix that is not positive definite, a non-positive shear modulus) are reported with MFH_ERR_INVALID when
 * the elements are embedded (next assembly / solve): the reference inverts blindly and fails later in CHOLMOD.    */
mfh_status mfh_material_const(mfh_ctx* ctx, const double* D);
mfh_status mfh_material_isotropic(mfh_ctx* ctx, double E, double nu);
mfh_status mfh_material_iso_field(mfh_ctx* ctx, const double* E, const double* nu);
mfh_status mfh_material_ortho_field(mfh_ctx* ctx, const double* params);
mfh_status mfh_material_tensor_field(mfh_ctx* ctx, const double* D);
/* element e's flattened D as the kernels use it (flatLen x flatLen) */
mfh_status mfh_material_get(mfh_ctx* ctx, int64_t elem, double* D);

/* ---------------------------------------------------------------- DoF map
 * == Simulator::applyPeriodicConditions' m_dofForNode (LinearElasticity.hh:825-854):
 * variable index = dim*dofForNode[node] + c.  NULL restores DoF(node)=node.                    */
mfh_status mfh_dof_map(mfh_ctx* ctx, const int32_t* dofForNode /* nNode or NULL */, int64_t nDoF);
/* PeriodicCondition (BoundaryConditions.hh:452-561): builds dofForNode from the bounding-box cell
 * with tolerance eps, installs it, flags internal boundary elements; returns nDoF.             */
mfh_status mfh_apply_periodic_conditions(mfh_ctx* ctx, double eps, int64_t* nDoF);
mfh_status mfh_get_dof_map(const mfh_ctx* ctx, int32_t* dofForNode /* nNode */, int64_t* nDoF);
/* The same on a ROW-PARTITIONED context (mfh_mesh_set; periodic cell problems across GPUs): the rank's local nodes map to local DoFs numbered
 * owned-first -- the rows of K are the first nOwnedDoF DoFs -- and the halo DoFs after them grouped by owner rank; mfh_dist_setup then takes its
 * lists in DoF numbers (send: owned DoFs, receive ranges: cover the halo DoFs). Every element that touches a node of an owned DoF must be local. */
mfh_status mfh_dof_map_partitioned(mfh_ctx* ctx, const int32_t* dofForNode /* nNode */, int64_t nDoF, int64_t nOwnedDoF);

/* ---------------------------------------------------------------- assembly
 * mfh_assemble == Simulator::m_assembleStiffnessMatrix (LinearElasticity.hh:1408-1466) followed
 * by TripletMatrix::sumRepeated + CSC build (SparseMatrices.hh:280-374,422-447): perElementStiffness
 * (LinearElasticity.hh:165-232) for every element, summed into a device-resident block-CSR K
 * (dim x dim blocks, FULL symmetric storage, structural zeros kept).                           */
mfh_status mfh_assemble(mfh_ctx* ctx, int32_t mode);
/* The symbolic phase alone (block-CSR pattern + gather lists; implied by mfh_assemble). It is the
 * hoisted, once-per-mesh part of sumRepeated's sort/merge (SparseMatrices.hh:280-374).          */
mfh_status mfh_symbolic(mfh_ctx* ctx, int32_t withScatterMap);
mfh_status mfh_symbolic_sizes(const mfh_ctx* ctx, int64_t* nChunk, int64_t* nContrib, int32_t* chunkSlots, int32_t* maxRowLen);
/* introspection of the symbolic structure (tests): any pointer may be NULL; the gather lists are
 * only retained on the host with option "keep_host_symbolic" (always in a host-only context)    */
mfh_status mfh_symbolic_get(const mfh_ctx* ctx, int32_t* rowPtr, int32_t* colIdx, int32_t* chunkRow, int64_t* contribPtr,
                            uint32_t* contribCode, uint16_t* contribSlot, int32_t* scatterSlot);
mfh_status mfh_matrix_info(const mfh_ctx* ctx, int64_t* nBlockRows, int64_t* nBlockCols, int64_t* nnzBlocks);
/* Storage of K (option "matrix_storage"): *upperOnly = 1 when only the blocks (r, c >= r) are stored and assembled -- the triangle the
 * reference's TripletMatrix holds --, *storedBlocks = their number. mfh_matrix_info and mfh_export_bsr describe K itself either way. */
mfh_status mfh_matrix_storage(const mfh_ctx* ctx, int32_t* upperOnly, int64_t* storedBlocks);
mfh_status mfh_export_bsr(mfh_ctx* ctx, int32_t* rowPtr /* nBlockRows+1 */, int32_t* colIdx /* nnzb */,
                          double* vals /* nnzb x dim*dim, row-major blocks */);
/* == TripletMatrix after m_assembleStiffnessMatrix + sumRepeated: upper triangle (row<=col) in
 * column-major sorted order with exact zeros pruned (SparseMatrices.hh:231-234,370-373); same
 * content as TripletMatrix::dumpBinary (SparseMatrices.hh:629-645). Call with i=j=v=NULL to get
 * a sufficient capacity in *nnz (the structural entries of the triangle); otherwise *nnz is the
 * capacity on entry and the number of entries written on return.                               */
mfh_status mfh_export_upper_triplets(mfh_ctx* ctx, uint64_t* i, uint64_t* j, double* v, uint64_t* nnz);
/* per-element dense Ke (debug/parity): full symmetric (n*dim)^2 row-major, local dof = dim*node+c */
mfh_status mfh_element_stiffness(mfh_ctx* ctx, int64_t firstElem, int64_t count, double* Ke);


/* SPSDSystem<Real>(K) for a CALLER-SUPPLIED symmetric positive (semi-)definite matrix (SparseMatrices.hh:2332-2348;
 * python binding sparse_matrices.SPSDSystem, sparse_matrices.cc:47-57): upper-triangle triplets, repeated entries
 * summed. Replaces any mesh in the context; afterwards mfh_fix_variables / mfh_solve / mfh_apply_K /
 * mfh_export_* work on it (one scalar variable per row, Jacobi-preconditioned CG). */
mfh_status mfh_matrix_set_upper_triplets(mfh_ctx* ctx, int64_t n, int64_t nnz, const uint64_t* i, const uint64_t* j, const double* v);
/* ---------------------------------------------------------------- constrained solve
 * mfh_fix_variables == SPSDSystem::fixVariables (SparseMatrices.hh:2389-2500): variables (scalar
 * indices dim*dof+c) pinned to values; K_rf u_f moves to the right-hand side (:2457-2470). Calls
 * accumulate like the reference; fixing a variable twice is an error ("Variable already fixed.").
 * mfh_solve == SPSDSystem::solve (SparseMatrices.hh:2515-2606) with CHOLMOD replaced by a HIP
 * preconditioned CG on the free variables; u holds fixed values at fixed variables (:2592-2605).
 * f/u: nrhs vectors of dim*nDoF doubles each.                                                  */
mfh_status mfh_clear_fixed(mfh_ctx* ctx);
mfh_status mfh_fix_variables(mfh_ctx* ctx, int64_t n, const int64_t* vars, const double* vals /* or NULL = 0 */);
mfh_status mfh_set_preconditioner(mfh_ctx* ctx, int32_t kind);
/* coarse-space facts of the last two-level setup: aggregates, coarse dimension, setup time (ms); note = why it fell back (or "") */
/* the preconditioner in use (for MFH_PRECOND_AUTO: the choice made for the mesh in hand -- made now if it has not been), whether it was chosen
 * automatically, and the stretch of the mesh the choice looked at (-1: not computed) */
mfh_status mfh_precond_choice(mfh_ctx* ctx, int32_t* kind, int32_t* isAuto, double* meshStretch);
mfh_status mfh_precond_info(const mfh_ctx* ctx, int32_t* nAggregates, int64_t* coarseDim, double* setup_ms, const char** note);
/* the hierarchy MFH_PRECOND_MULTIGRID built at the last solve: DoFs of the quadratic and of the linear level, the largest eigenvalues
 * of their Jacobi-preconditioned operators (the Chebyshev smoothers' upper bounds), setup time; zeros when it is not in use. On
 * linear elements there is no quadratic level: fineDoF == coarseDoF and lambdaMaxFine == 0. */
mfh_status mfh_multigrid_info(const mfh_ctx* ctx, int64_t* fineDoF, int64_t* coarseDoF, double* lambdaMaxFine, double* lambdaMaxCoarse,
                              double* setup_ms);
mfh_status mfh_solve(mfh_ctx* ctx, int32_t nrhs, const double* f, double* u,
                     double rtol, int32_t maxit, mfh_solve_info* info /* ONE entry: the last right-hand side */);
/* the same with one mfh_solve_info per right-hand side. Where batches pay, right-hand sides are solved in batches (3D: 6 or 2, 2D: 3), like
 * the reference factors once and back-substitutes per right-hand side (PeriodicHomogenization.hh:34-54): under MFH_PRECOND_MULTIGRID the
 * coarse levels of every V-cycle serve the whole batch (option "mg_batch", default on); the Chronopoulos-Gear batches of the other
 * preconditioners (option "batch_rhs") are off by default (measured slower). info[k].reserved = size of the batch rhs k was solved in;
 * info[k].solve_ms of a batch is the batch's device time. */
mfh_status mfh_solve_batch(mfh_ctx* ctx, int32_t nrhs, const double* f, double* u,
                           double rtol, int32_t maxit, mfh_solve_info* info /* nrhs entries */);
/* == Simulator::applyStiffnessMatrix (LinearElasticity.hh:801-823), using the assembled K       */
mfh_status mfh_apply_K(mfh_ctx* ctx, const double* u /* dim*nDoF */, double* Ku);

/* ---------------------------------------------------------------- Simulator-level helpers
 * Box regions are inclusive (BBox::containsPoint, Geometry.hh:276-279).
 * mfh_bc_dirichlet_box == DirichletCondition branch of applyBoundaryConditions (:939-949): every
 *   BOUNDARY NODE whose position lies in the box gets `value` on the components in compMask
 *   (bit c = component c); conflicting values (>1e-10) fail (LinearElasticity.hh:386-406).
 * mfh_bc_neumann_box == NeumannCondition branch (:897-933): every BOUNDARY ELEMENT whose vertex
 *   barycentre lies in the box; FORCE divides by the region area; PRESSURE uses value[0].
 * relative != 0 interprets the corners relative to the mesh bounding box ("box%",
 *   BoundaryConditions.cc:310-316).                                                             */
mfh_status mfh_bc_clear(mfh_ctx* ctx);
mfh_status mfh_bc_dirichlet_box(mfh_ctx* ctx, const double* minCorner, const double* maxCorner,
                                int32_t relative, const double* value, int32_t compMask);
mfh_status mfh_bc_neumann_box(mfh_ctx* ctx, const double* minCorner, const double* maxCorner,
                              int32_t relative, const double* value, int32_t kind);
/* == DirichletNodesCondition branch (:991-1002) and the carrier of expression-valued Dirichlet regions (ExpressionVector
 * components are evaluated per node by the caller): per-node displacement values [n x dim]; "Condition applied to
 * non-boundary node i" and "Conflicting dirichlet displacements." as in the reference. */
mfh_status mfh_bc_dirichlet_nodes(mfh_ctx* ctx, int64_t n, const int64_t* nodes, const double* values, int32_t componentMask);
/* == NeumannElementsCondition branch (:966-990) after matching and the force / region-area division, and the carrier of
 * expression-valued traction regions: sets the traction [n x dim] of the listed boundary elements. */
mfh_status mfh_bc_neumann_elements(mfh_ctx* ctx, int64_t n, const int64_t* bdryElems, const double* tractions);
mfh_status mfh_bc_delta_force(mfh_ctx* ctx, int64_t node, const double* force);
/* == m_getDirichletVarsAndValues (:1469-1518). vars==NULL returns the count.                   */
mfh_status mfh_bc_dirichlet_vars(mfh_ctx* ctx, int64_t* vars, double* vals, int64_t* n);
/* == m_pinNode (:1595-1618): first non-boundary node, else node 0.                             */
mfh_status mfh_pin_node(const mfh_ctx* ctx, int64_t* node);
/* == Simulator::neumannLoad (:703-717); out: dim*nDoF                                          */
mfh_status mfh_neumann_load(mfh_ctx* ctx, double* out);
/* == Simulator::constantStrainLoad (:551-562, :135-162); cstrain flattened (flatLen, TENSOR shear) */
mfh_status mfh_constant_strain_load(mfh_ctx* ctx, const double* cstrainFlat, double* out);
/* == Simulator::solve(f) (:479-487 -> m_buildConstrainedSystem :1377-1404): assembles if needed,
 * fixes the Dirichlet variables (+ the pin node if usePin), PCG-solves, returns dofToNodeField
 * (:664-677) as nNode x dim. f==NULL uses neumannLoad() (:657).                                */
mfh_status mfh_sim_solve(mfh_ctx* ctx, const double* f /* dim*nDoF or NULL */, int32_t usePin,
                         double* uNodes /* nNode x dim */, double rtol, int32_t maxit, mfh_solve_info* info);
/* Simulator::solve with ALL of assembleConstrainedSystem (LinearElasticity.hh:1201-1249). flags:
 *   MFH_SOLVE_PIN               m_useNRTPinConstraint: translations removed by pinning a node (:1595-1618) instead of rows
 *   MFH_SOLVE_NO_RIGID_MOTION   m_useRigidMotionConstraint (`no_rigid_motion` in .bc files): rotation rows (:1525-1566,
 *                               skipped under periodic conditions) + translation rows or pin; rigidMotionRHS optional
 *   MFH_SOLVE_ALLOW_ILL_POSED   allowIllPosed: Dirichlet variables only
 *   0                           default: analyzeDirichletPosedness (:1169-1190) adds translation constraints for the
 *                               components without any Dirichlet condition; "Unimplemented" if nothing is constrained.
 * The reference hands the resulting KKT system to UMFPACK; here the <= 6 constraint rows are eliminated around SPD
 * PCG solves (one consistent singular solve when they exactly remove the rigid motions, k + 1 solves otherwise). */
enum { MFH_SOLVE_PIN = 1, MFH_SOLVE_NO_RIGID_MOTION = 2, MFH_SOLVE_ALLOW_ILL_POSED = 4 };
mfh_status mfh_sim_solve_constrained(mfh_ctx* ctx, const double* f, int32_t flags, const double* rigidMotionRHS,
                                     int32_t nRigidRHS, double* uNodes, double rtol, int32_t maxit, mfh_solve_info* info);
/* Simulator::solve for nrhs load vectors on ONE constrained system -- what solveCellProblems does with its 3 / 6 constantStrainLoad vectors
 * (PeriodicHomogenization.hh:34-54: one Simulator, one factorisation, one back-substitution per load). f: nrhs x dim*nDoF, uNodes:
 * nrhs x nNode*dim, info: nrhs entries. Where the constraints leave a positive definite system (Dirichlet conditions, or periodic conditions
 * with the pinned node) the right-hand sides go through mfh_solve_batch's batches -- under MFH_PRECOND_MULTIGRID the linear / aggregate /
 * dense levels of every V-cycle are run once for the whole batch (option "mg_batch") --; systems with constraint rows are solved one
 * right-hand side after the other as by mfh_sim_solve_constrained. */
mfh_status mfh_sim_solve_batch(mfh_ctx* ctx, int32_t nrhs, const double* f, int32_t flags, double* uNodes, double rtol, int32_t maxit,
                               mfh_solve_info* info);
/* == solveCellProblems' loop (PeriodicHomogenization.hh:47-53): w[k] = Simulator::solve(constantStrainLoad(cstrains[k])) for k < nStrains on the
 * system the context's conditions describe (the caller has applied the periodic conditions; flags as for mfh_sim_solve_constrained:
 * MFH_SOLVE_PIN | MFH_SOLVE_NO_RIGID_MOTION is solveCellProblems' configuration). cstrains: nStrains x flatLen, flattened with TENSOR shear
 * like mfh_constant_strain_load's argument (the caller passes -e_ij); wNodes: nStrains x nNode*dim. Under MFH_PRECOND_MULTIGRID on a quadratic
 * mesh nothing but the solutions crosses the PCIe bus: the load vectors are formed on the device through the matrix-free operator's lists
 * (the element routine with u = 0 and the constant strain added), the right-hand sides share the coarse levels of every V-cycle (option
 * "mg_batch"), dofToNodeField runs on the device. Every other configuration takes the general route (host load vectors, mfh_sim_solve_batch). */
mfh_status mfh_solve_cell_problems(mfh_ctx* ctx, int32_t nStrains, const double* cstrains, int32_t flags, double* wNodes, double rtol,
                                   int32_t maxit, mfh_solve_info* info);
/* == averageStrainField / averageStressField (:528-549, :99-123): per element, flattened (flatLen) */
mfh_status mfh_average_strain(mfh_ctx* ctx, const double* uNodes, double* strain /* nElem x flatLen */);
mfh_status mfh_average_stress(mfh_ctx* ctx, const double* uNodes, double* stress /* nElem x flatLen */);
/* sum_e vol_e C_e : (averageStrain_e(u) + cstrain)  [flatLen values]: the element loop of homogenizedElasticityTensor
 * (PeriodicHomogenization.hh:72-100: Eh.DRow(i) = 1/|Y| sum_e vol_e [E_e : strain(w_i) + E_e.DRow(i)]) reduced on the device; cstrainFlat
 * (flattened, tensor shear entries; NULL = none) stands for the affine displacement of the probe strain e_i. */
mfh_status mfh_integrated_stress(mfh_ctx* ctx, const double* uNodes, const double* cstrainFlat, double* out /* flatLen */);
/* == strainField / stressField, elementStrain / elementStress (:493-526; Element::strain :99-117): the nodal values of the
 * degree-(Deg-1) strain interpolant of every element -- one value for P1, the values at the dim+1 corners for P2 (edge
 * nodes of an upsampled field are the means of their end points). out: [nElem][1 | dim+1][flatLen]. */
mfh_status mfh_strain_field(mfh_ctx* ctx, const double* uNodes, int32_t wantStress, double* out);
/* The same interpolant restricted to the boundary elements (restrictInterpolant, InterpolantRestriction.hh:29-66, as used
 * for the boundary stresses of PeriodicHomogenization.hh:316-340 and the integrand of homogenizedElasticityTensorGradient
 * :226-288): values of the parent element's strain at the boundary element's corners, in the boundary element's vertex
 * order. out: [nBE][1 | dim][flatLen]. */
mfh_status mfh_boundary_strain_field(mfh_ctx* ctx, const double* uNodes, int32_t wantStress, double* out);

/* ---- discrete shape derivatives, forward mode (LinearElasticity.hh:234-330 at element level; Simulator level
 * :1297-1374). deltaP is a per-vertex perturbation field [nVert x dim] (indexed by the node id of the element corners:
 * vertex nodes come first in the FEM numbering); nodal fields are held fixed. Each is the original device kernel
 * evaluated on perturbed barycentric gradients (delta grad lambda, EmbeddedElement.hh:269-278; delta vol/vol, :366-372).
 *   mfh_apply_delta_K               == Simulator::applyDeltaStiffnessMatrix(u, deltaP) (:1301-1328): (delta K) u,
 *                                      u per NODE, result per DoF; deltaPerElementStiffness (:306-330) is never formed
 *   mfh_delta_constant_strain_load  == Simulator::deltaConstantStrainLoad(cstrain, deltaP) (:1331-1348)
 *   mfh_delta_average_strain        == Simulator::deltaAverageStrainField(u, deltaU, deltaP) (:1364-1374):
 *                                      (delta strain)(u) + strain(deltaU); wantStress != 0 contracts with C (deltaStress :280-286)
 *   mfh_mutual_energies             sum_e int (e^ij + eps(w^ij)) : C : (e^kl + eps(w^kl)) dV as a flatLen x flatLen
 *                                      matrix (= |Y| Ch, energy form of PeriodicHomogenization.hh:146-186); with deltaP != NULL
 *                                      its discrete shape derivative in the volume form of PeriodicHomogenization.hh:484-491
 *                                      (what deltaHomogenizedElasticityTensor :492-514 evaluates through boundary integrals).
 *                                      w: [flatLen][nNode][dim] per-node fluctuation displacements. */
mfh_status mfh_apply_delta_K(mfh_ctx* ctx, const double* uNodes, const double* deltaP, double* out /* dim*nDoF */);
mfh_status mfh_delta_constant_strain_load(mfh_ctx* ctx, const double* cstrainFlat, const double* deltaP, double* out /* dim*nDoF */);
mfh_status mfh_delta_average_strain(mfh_ctx* ctx, const double* uNodes, const double* deltaU, const double* deltaP,
                                    int32_t wantStress, double* out /* nElem x flatLen */);
mfh_status mfh_mutual_energies(mfh_ctx* ctx, const double* w, const double* deltaP /* or NULL */, double* out /* flatLen^2 */);
/* == homogenizedElasticityTensorDiscreteDifferential (PeriodicHomogenization.hh:372-480) before its division by |Y|:
 * the exact derivative of every mutual energy with respect to every vertex coordinate (reverse mode: all directions in
 * one element sweep). out: [pair][nVert][dim], pair = row-major index over the upper triangle ij <= kl of the
 * flatLen x flatLen tensor (flatLen (flatLen + 1) / 2 pairs). Contracting it with a perturbation field reproduces
 * mfh_mutual_energies(w, deltaP). */
mfh_status mfh_mutual_energy_differential(mfh_ctx* ctx, const double* w, double* out);
/* Select the operator (default MFH_OP_ELASTICITY). Keeps mesh, DoF map, pattern and gather lists; drops the
 * assembled values and the fixed variables (their numbering depends on the block size). */
mfh_status mfh_set_operator(mfh_ctx* ctx, int32_t op);
/* PoissonMesh::gradUAverage (Poisson.hh:121-131): per-element average gradient of a scalar nodal field */
mfh_status mfh_average_gradient(mfh_ctx* ctx, const double* uNodes /* nNode */, double* grad /* nElem x dim */);

/* ---------------------------------------------------------------- multi-GPU solve (one process per GPU)
 * The reference is single-process (TBB, Parallelism.hh:31-43): these entry points have no counterpart to cite beyond the
 * serial path they parallelise (Simulator::solve, LinearElasticity.hh:479-487; SPSDSystem::solve, SparseMatrices.hh:2515-2606).
 * Rows (nodes) are partitioned: every rank builds its context with mfh_mesh_set(..., nOwned) -- owned nodes first, halo
 * nodes after them GROUPED BY OWNER RANK -- and assembles its rows without communication. The solve needs, per PCG
 * iteration, one neighbour exchange of the halo entries and ONE all-reduce of 4 nrhs doubles (two with the two-level
 * preconditioner); both go through an mfh_comm:
 *   mfh_comm_create_rccl       RCCL communicator of `world` ranks created from a shared unique id (rank 0 calls
 *                              mfh_rccl_get_unique_id and distributes the 128 bytes by any means: MPI, a file, torch's store).
 *                              RCCL is looked up with dlopen (the process's own copy if it has one, e.g. PyTorch's), so the
 *                              library links against no particular communication stack.
 *   mfh_comm_create_callbacks  the two collectives supplied by the caller (MPI, torch.distributed/gloo in the CPU-staged
 *                              tests ...). Buffers are DEVICE pointers; the call must be ordered on `hipStream` (enqueue
 *                              there, or synchronise the stream, do the transfer and return).
 * mfh_dist_setup: peers = neighbouring ranks; sendNodes[sendPtr[k] .. sendPtr[k+1]) = OWNED local nodes whose values rank
 *   peers[k] reads (in the order that rank stores them); halo nodes nOwned + [recvPtr[k], recvPtr[k+1]) are owned by peers[k].
 * mfh_dist_two_level: the two-level preconditioner with GLOBAL aggregates (arguments as mfh_tl_partitioned_begin; the
 *   coarse operator is summed over the ranks, every rank inverts it redundantly).
 * mfh_dist_solve: f / u are nrhs vectors of dim * nOwned doubles (this rank's rows). Fixed variables (mfh_fix_variables,
 *   local numbering) must be given for halo nodes as well as owned ones.                                              */
typedef struct mfh_comm mfh_comm;
typedef struct mfh_rccl_unique_id { char internal[128]; } mfh_rccl_unique_id;
typedef mfh_status (*mfh_allreduce_fn)(void* user, double* devBuf, int64_t n, void* hipStream);
typedef mfh_status (*mfh_exchange_fn)(void* user, int32_t nPeers, const int32_t* peers, const double* const* sendBufs,
                                      const int64_t* sendCounts, double* const* recvBufs, const int64_t* recvCounts, void* hipStream);
mfh_status  mfh_rccl_get_unique_id(mfh_rccl_unique_id* id);
mfh_status  mfh_comm_create_rccl(mfh_ctx* ctx, const mfh_rccl_unique_id* id, int32_t rank, int32_t world, mfh_comm** out);
mfh_status  mfh_comm_create_callbacks(int32_t rank, int32_t world, void* user, mfh_allreduce_fn allreduce_sum,
                                      mfh_exchange_fn exchange, mfh_comm** out);
void        mfh_comm_destroy(mfh_comm* comm);
const char* mfh_comm_describe(const mfh_comm* comm);
/* in-place sum of n doubles in device memory over the ranks (blocking) */
mfh_status  mfh_comm_allreduce(mfh_ctx* ctx, mfh_comm* comm, double* devBuf, int64_t n);
/* ring shift of a known message + all-reduces with known sums (several rounds when peer transfers are enabled): checks the transport
 * in use end to end */
mfh_status  mfh_comm_selftest(mfh_ctx* ctx, mfh_comm* comm);
/* First contact with the devices and links of a node, before any solve (collective): device memory head-room, the devices of all ranks
 * and hipDeviceCanAccessPeer towards each, which IPC slabs are mapped (after mfh_comm_enable_peer), an all-reduce of ones, and the
 * bandwidth of a ring of messageBytes-long messages (send to rank + 1 while receiving from rank - 1) on the transport underneath and
 * through the peer transfers. Layout of `out` (>= 16 + 2 world doubles): meshfem_amd/csrc/mfh_solver.cpp, mfh_comm_preflight. */
mfh_status  mfh_comm_preflight(mfh_ctx* ctx, mfh_comm* comm, int64_t messageBytes, double* out, int64_t nOut);
/* Direct device-to-device transfers for the ranks of ONE node, layered on an existing communicator (collective; the HIP IPC handles
 * travel through the communicator's own all-reduce): every rank exports one slab of fine-grained device memory; the halo exchange of
 * the PCG then writes straight into the neighbours' slabs (stores over xGMI, one release flag per message, two buffers per pair) and
 * all-reduces of up to 2^21 doubles (the dot products: ONE single-workgroup kernel; the replicated coarse levels) are all-to-all
 * writes summed in rank order -- no library call and no host code per iteration, bit-identical sums on every rank. Larger
 * all-reduces and halo widths the staging was not sized for keep using the communicator underneath. The ranks must be separate
 * processes (several may share a device); needs HSA_ENABLE_IPC_MODE_LEGACY=0 on hosts whose driver only supports dmabuf IPC.
 * A rank that waits longer than MFH_PEER_TIMEOUT_S (60) seconds for a neighbour gives up and the solve returns MFH_ERR_HIP: the
 * device is never left spinning. mfh_comm_selftest exercises the path; mfh_comm_disable_peer (collective) returns to the transport
 * underneath. A communicator serves ONE host thread and one collective at a time (like a context): the message numbers and staging
 * buffers of a pair are shared by everything that runs on it -- two contexts may name the same communicator in mfh_dist_setup, their
 * solves must not overlap. */
mfh_status  mfh_comm_enable_peer(mfh_ctx* ctx, mfh_comm* comm);
mfh_status  mfh_comm_disable_peer(mfh_ctx* ctx, mfh_comm* comm);
mfh_status  mfh_dist_setup(mfh_ctx* ctx, mfh_comm* comm, int32_t nPeers, const int32_t* peers, const int64_t* sendPtr /* nPeers+1 */,
                           const int32_t* sendNodes, const int64_t* recvPtr /* nPeers+1 */);
mfh_status  mfh_dist_two_level(mfh_ctx* ctx, int32_t nAgg, const int32_t* aggOfNode, const double* relPos);
mfh_status  mfh_dist_solve(mfh_ctx* ctx, int32_t nrhs, const double* fOwned, double* uOwned, double rtol, int32_t maxit,
                           mfh_solve_info* info /* nrhs entries, or NULL */);
/* (K u) on this rank's rows for a field given on its rows (halo entries fetched from the owners) */
mfh_status  mfh_dist_apply_K(mfh_ctx* ctx, const double* uOwned, double* KuOwned);
/* What the last mfh_dist_solve did on this rank. transport: what carried the halo exchange (1 RCCL send/recv, 2 peer copies over HIP
 * IPC, 3 caller callbacks). With option "dist_profile" 1 the first operator applications of a solve are bracketed by HIP events:
 * exchange_ms (pack done -> halo arrived, communication stream), interior_ms (the element blocks / row chunks that read no halo column),
 * exposed_wait_ms (how long the compute stream then still waited for the halo: 0 = the exchange is hidden), boundary_ms (the blocks
 * that read the halo), operator_ms (all of it), averages over profiled_applications. */
typedef struct mfh_dist_stats {
    int32_t world, rank, transport, peer_enabled;
    int64_t halo_nodes_sent, halo_nodes_received, halo_bytes_per_exchange /* sent + received, one right-hand side */;
    int64_t interior_items, boundary_items;
    double  exchange_ms, interior_ms, boundary_ms, exposed_wait_ms, operator_ms;
    int32_t profiled_applications, reserved;
    int64_t exchanges;                                   /* since mfh_dist_setup */
    int64_t peer_halo_messages, peer_halo_bytes;         /* through the peer transfers, since they were enabled */
    int64_t allreduces_small, allreduces_large, fallback_exchanges, fallback_allreduces;
} mfh_dist_stats;
mfh_status  mfh_dist_get_stats(mfh_ctx* ctx, mfh_dist_stats* out);
/* blocking copy helper for callback communicators that stage through the host: kind 0 host->device, 1 device->host,
 * 2 device->device; hipStream NULL = the context's stream */
mfh_status  mfh_dev_memcpy(mfh_ctx* ctx, void* dst, const void* src, int64_t bytes, int32_t kind, void* hipStream);

/* Device memory. The library allocates from a per-process ARENA (meshfem_amd/csrc/mfh_pool.cpp): released device memory is not returned to
 * the driver but kept in the hipMalloc segments it came in, where free neighbours are merged and any later request is cut from the smallest
 * free chunk that holds it; the driver is asked for a new segment only when nothing fits. Reason: on ROCm 7.2 / MI355X a hipMalloc that
 * follows large hipFree calls of the same process takes seconds (profiles/r04_malloc_probe.txt), and the setup phases of a context allocate
 * and release several times the memory they keep. The reference reserves its triplet storage once (LinearElasticity.hh:1441-1443).
 * Bounds: free bytes kept while contexts are alive <= MFH_DEVICE_CACHE_MB (default half of the device's memory; 0 = plain hipMalloc /
 * hipFree); when the LAST context of a device closes the arena is trimmed to at most MFH_DEVICE_CACHE_IDLE_MB of free memory (default a
 * quarter of the device), so that other allocators of the process (torch, RCCL) find the rest. A request the driver cannot serve releases every free segment and is repeated. mfh_device_cache_trim returns every free segment to
 * the driver at once. mfh_device_arena_stats: out8 = {bytes held, live bytes, live high-water mark, segments, free chunks, bytes returned
 * to the driver so far, bytes waiting for a device-wide synchronisation, bound on the free bytes}.
 * Debug aid: MFH_ARENA_GUARD=1 puts 512 pattern bytes behind every buffer and checks them when the buffer is released -- a kernel that wrote past
 * the end of its buffer aborts the process with the buffer's size (synchronous fills and read-backs: for test runs; scripts/r06/run_guard.sh). */
/* mfh_device_reserve: `bytes` taken from the driver now and kept as free space of the arena -- the reference's "reserve once"
 * (LinearElasticity.hh:1441-1443) for a caller that knows roughly what its mesh will need (mfh_context_bytes_estimate; mfh_device_reserve_for
 * does both). Contexts created afterwards cut their buffers from it: no call to the driver during their setup. With async != 0 the call returns
 * at once and the allocation runs on threads of its own (e.g. while the caller reads its mesh); the library's next allocation that finds nothing
 * waits for it. Worth it on a device nobody has used since boot, where the driver clears the memory a process takes beyond the first ~66 GB
 * while it is being allocated (profiles/r05_large_allocation_trace_119.log).
 * The reservation is made in TWO segments: one for the value array of K (its share of the request: mfh_device_reserve_for knows it from the mesh
 * kind, mfh_device_reserve assumes the 43 % of a quadratic 3D context) and one for everything else -- the arena never puts anything else into
 * the value array's segment (and gives the array a segment of its own when nothing was reserved): with everything in ONE hipMalloc the values and
 * all other buffers lie in one physical run, and the assembly kernel then runs at the slow end of its placement spread (3.27 against 2.9-3.1 ms at
 * 5 M quadratic tets, 24.4 against 22.6 ms at 40 M; docs/design/04_2_k_assemble_gather.md (xi)). A synchronous reservation the driver cannot
 * serve returns MFH_ERR_HIP. */
mfh_status mfh_context_bytes_estimate(int32_t dim, int32_t deg, int64_t nElem, int64_t* totalBytes, int64_t* kValueBytes);
mfh_status mfh_device_reserve_for(int32_t device, int32_t dim, int32_t deg, int64_t nElem, int32_t async);
mfh_status mfh_device_reserve(int32_t device, int64_t bytes, int32_t async);
/* option "placement_trials": kernel time (ms) of every candidate of the last trials, the first being the buffer of the symbolic phase; *n = how many */
mfh_status mfh_placement_info(const mfh_ctx* ctx, int32_t cap, double* ms, int32_t* n);
mfh_status mfh_device_cache_trim(void);
mfh_status mfh_device_cache_stats(int32_t device, int64_t* cachedBytes, int64_t* blocks, int64_t* hits, int64_t* misses, int64_t* flushes);
mfh_status mfh_device_arena_stats(int32_t device, int64_t* out8);

/* ---------------------------------------------------------------- introspection, options
 * (kernel timers, operator statistics, test hooks and the device-pointer building blocks: include/meshfem_hip_extras.h) */
mfh_status mfh_get_timing(const mfh_ctx* ctx, mfh_timing* out);
/* option knobs (string key, numeric value): "chunk_slots", "contrib_order" (0 rank-major, 1 element-major,
 * 2 slot-major), "check_every", "keep_host_symbolic", "agg_nodes" (target DoFs per aggregate of the two-level
 * preconditioner), "reembed" (1: every mfh_assemble re-runs the element-embedding kernel as well),
 * "matrix_free" (operator of mfh_solve / mfh_apply_K / mfh_dev_spmv: 0 = assembled block-CSR SpMV, 1 = matrix-free
 *   (k_mf_cluster + k_mf_rows: element stresses recomputed, K not read), -1 = auto (default): matrix-free for elasticity
 *   (quadratic: 6x faster than the assembled SpMV; linear: 1.2x at 1 M tets, 1.8x from 6 M on), assembled for the scalar operators; both give
 *   K x up to rounding),
 * "matrix_free_mode" (4 default: cluster variant, forces of a block of 512-1024 consecutive elements summed in LDS | 3 two-pass, forces in
 *   list order | 2 two-pass, forces element-major | 1 per-pair block evaluation), "mf_chunk_rows", "mf_chunk_pairs",
 * "mf_geometry_from_vertices" (1 default: with a constant material the matrix-free operator recomputes the element gradients from the
 *   corner positions instead of reading the element records), "mf_xcd_group" (32 default: consecutive element blocks per XCD),
 *   "mf_lane_stride" (37 default: lane-to-element stride inside a block, against same-address LDS atomics),
 * "pcg_graph" (1 default: blocks of check_every PCG iterations are replayed from a hipGraph),
 * "refine" (1 default: when a converged solve of mfh_solve / mfh_sim_solve* ends with a TRUE residual ||f - K u|| / ||b|| above twice the
 *   tolerance -- the recurrence residual drifts over thousands of iterations -- the correction K du = f - K u is solved to what is missing
 *   and added, up to three times: the answer is as good as the tolerance says, like a direct solver's; solves that end within it are untouched),
 * "matrix_storage" (which blocks of K are stored and assembled. 1: only the blocks (r, c >= r), the triangle the reference's TripletMatrix
 *   holds -- half the bytes and block arithmetic of the assembly; serves mfh_export_upper_triplets, mfh_export_bsr (mirrored on the host),
 *   the (block-)Jacobi and two-level PCG on the matrix-free operator, and mfh_apply_K / mfh_dev_spmv through k_spmv_sym (the transposed half
 *   of the product added with global atomics: 2.5x slower than the product from both triangles); the PCG on the assembled SpMV and the probing
 *   construction of the coarse operator return MFH_ERR_UNSUPPORTED on it. 0: both triangles. -1 default: automatic -- the upper triangle exactly when nothing
 *   multiplies by the stored K (elasticity on the matrix-free operator), both triangles otherwise; changing an option that
 *   decides this re-runs the symbolic phase on the next use. mfh_matrix_info / mfh_export_bsr describe K itself either way,
 *   mfh_matrix_storage what is stored),
 * "placement_trials" (0 default; N <= 8: at the first gather-mode assembly after a symbolic phase the buffer of the K values is allocated up to N more
 *   times, the assembly kernel timed on each (two passes), and the fastest kept -- where the driver puts these bytes moves the kernel between
 *   2.9 and 3.3 ms at 5 M quadratic tets, docs/design/04_2_k_assemble_gather.md (xi); costs ~10 ms and one more buffer of the size of K per trial,
 *   once per symbolic phase: for callers that assemble hundreds of times on one mesh. mfh_placement_info reads the candidates' times),
 * "pcg_variant" (1: Chronopoulos-Gear PCG, one reduction point per iteration and one fused vector kernel -- always used by
 *   mfh_dist_solve and for batches; 0: the classic two-reduction PCG, one right-hand side at a time; -1 default: classic for
 *   a single right-hand side on an unpartitioned context, where it is 6-14 % faster per iteration), "dist_pcg_variant"
 *   (mfh_dist_solve with one right-hand side: 1 default = Chronopoulos-Gear, ONE all-reduce per iteration; 0 = classic loop, two
 *   all-reduces and one vector pass less -- which is faster depends on the node's all-reduce latency), "batch_rhs" (0 default:
 *   1 solves several right-hand sides per operator pass of the Chronopoulos-Gear loop, see mfh_solve_batch; block-Jacobi / two-level only),
 * "mg_batch" (1 default: several right-hand sides under MFH_PRECOND_MULTIGRID on an unpartitioned quadratic context run one classic PCG loop
 *   each, in lockstep, and share the linear / aggregate / dense levels of every V-cycle; 0: one right-hand side at a time),
 * "mg_fuse" (1 default: with one Chebyshev step on the quadratic level of an unpartitioned hierarchy, the PCG loop's residual update also writes
 *   the V-cycle's pre-smoothed start and the cycle's last smoothing step also forms r.z -- two kernels and three vector passes less per
 *   iteration, the same iterates; 0: separate kernels),
 * "mg_dinv_fp32" (1 default: those two fused kernels read an FP32 copy of the quadratic level's inverse diagonal blocks -- a third of their traffic;
 *   the smoother only, both steps of a cycle the same copy; 0: the FP64 blocks),
 * "solve_homogeneous" (1: mfh_solve treats the fixed variables as fixed to ZERO whatever values were given -- the
 *   homogeneous solves K y = C^T of a Schur-complement elimination of constraint rows, SparseMatrices.hh:2572-2590),
 * "periodic_ignore_mismatch" (1: nodes of a periodic face without a partner keep their own DoF, PeriodicCondition's
 *   ignoreMismatch / matchPermittingMismatch; 0 default: a mismatch is an error like PeriodicBoundaryMatcher::match),
 * "periodic_ignore_dims" (bit a set: dimension a is NOT periodic, PeriodicCondition's ignoreDims) -- both read by the next
 *   mfh_apply_periodic_conditions,
 * "mg_steps_fine" / "mg_steps_coarse" (1 / 1: Chebyshev steps of MFH_PRECOND_MULTIGRID before and after the coarse correction on the quadratic /
 *   linear level), "mg_ratio_fine" / "mg_ratio_coarse" (0.3 / 0.3: the smoothers act on [ratio lambda_max, lambda_max]), "mg_coarse_cycles"
 *   (1: cycles of the linear level per application), "mg_eig_margin" (1.1: factor on the power-iteration estimates), "mg_agg_nodes",
 * "mg_coarse_fp32" (1 default: INSIDE the multigrid preconditioner the assembled matrix of the linear level and the stencil operators of the
 *   aggregate levels are read from FP32 copies -- products, sums and every vector stay FP64, and so do K, K x and the residuals of the PCG itself;
 *   the preconditioner is a fixed SPD operator either way, the solution is the FP64 one; 0: FP64 storage throughout),
 * "asm_packed_codes" (1 default: the device copy of the gather lists is chunk-relative and packed, see k_assemble_gather),
 * "asm_chunk_order" (0 default; 1: the assembly visits the row chunks in the order of the elements they gather from),
 * "deterministic" (1: run-to-run BIT-REPRODUCIBLE assembly, operator and PCG -- the counterpart of the reference's serial, reproducible
 *   scatter into the triplet list (LinearElasticity.hh:1454-1455; SparseMatrices.hh:288,319-324 switches its one unordered loop off). The
 *   default kernels accumulate with LDS / global atomics in the order the waves happen to arrive, so the last bits of K, K x and the dot
 *   products differ between two runs; with the option the waves of a workgroup add in wave order (assembly, interface rows of the
 *   operator) or into accumulators of their own summed in wave order (element blocks of the operator), every dot product goes through a
 *   fixed two-stage tree, and the PCG runs the classic loop. Two assemblies give identical bits, two solves identical displacements.
 *   Every preconditioner (the Galerkin products of the coarse levels keep one accumulator table per wave, the transfers between the
 *   aggregate levels gather); the global-atomic assembly variant and matrix_free_mode 1 - 3 are refused. Measured cost: bench.py
 *   variants.deterministic),
 * "dist_profile" (1: time the halo exchange against the interior work in the first operator applications of every mfh_dist_solve,
 *   see mfh_dist_get_stats),
 * "symbolic_device", "topology_device", "tl_probe", "tl_host_inverse" (validation variants of setup phases) */
mfh_status mfh_set_option(mfh_ctx* ctx, const char* key, double value);
#ifdef __cplusplus
}
#endif
#endif /* MESHFEM_HIP_H */
