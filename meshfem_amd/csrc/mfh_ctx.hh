// Context object behind the opaque mfh_ctx handle and the internal helpers shared by the translation units that implement
// the C ABI (mfh_api.cpp: context, mesh, materials, assembly, solver, device-pointer building blocks, measurement;
// mfh_simulator.cpp: boundary conditions, loads, Simulator-level solve with constraint rows, strain / stress fields,
// shape derivatives).
#pragma once
#include "mfh_internal.hh"
#include <cmath>
#include <map>
#include <mutex>

using namespace mfh;   // private header of the library's own translation units

struct mfh_ctx {
    int device = 0;
    bool hostOnly = false;            // device == -1: mesh/symbolic host logic only (CPU tests)
    bool hierarchyLevel = false;      // the linear level of another context's multigrid hierarchy (mfh_multigrid.cpp)
    bool keepHostSymbolic = false;
    int xcdSwizzle = 0;              // option "xcd_swizzle": XCD-contiguous work mapping in the chunked kernels (measured SLOWER
                                     // than the round-robin default on MI355X: DESIGN.md section 4.8)
    RawVec<double> hLoad, hX, hXBatch, hLoadBatch;   // host scratch of Simulator::solve (kept between solves; sized with resize_prefaulted: a value-initialising
                                     // resize of 1.4 GB at 119^3 is 0.25 s of single-threaded page faults, twice per first solve)
    bool alwaysReembed = false;      // option "reembed": every mfh_assemble re-runs the embedding kernel
    // option "deterministic": run-to-run bit-reproducible assembly, operator and PCG. The reference's scatter into the triplet list is serial
    // and therefore reproducible (LinearElasticity.hh:1454-1455; the one place that is not, it documents and switches off,
    // SparseMatrices.hh:288,319-324). Here the default kernels accumulate with LDS / global atomics in whatever order the waves arrive; the
    // option orders them: the waves of a workgroup add one after the other (k_assemble_gather, k_mf_rows) or into arrays of their own
    // (k_mf_cluster, k_tl_rap_agg), the dot products go through a fixed two-stage tree (commit_sums), the PCG runs the classic loop; the
    // transfers and Galerkin products between the aggregate levels of the multigrid hierarchy gather (no atomics) in every mode. All
    // preconditioners; contexts that own all their rows and row-partitioned ones (whose all-reduce is in rank order with the peer transfers).
    bool deterministic = false;
    DBuf<double> detPartials;
    DBuf<unsigned> detCounter;
    bool periodicIgnoreMismatch = false;   // option "periodic_ignore_mismatch": PeriodicCondition(..., ignoreMismatch)
    int periodicIgnoreDims = 0;            // option "periodic_ignore_dims": bit a set = dimension a is not periodic
    hipStream_t stream = nullptr;
    bool ownStream = true;
    bool arenaCounted = false;          // counted by the device arena (mfh_create of a device context): mfh_destroy reports the close
    std::string err;

    // ---- mesh
    HostMesh mesh;
    bool haveMesh = false;
    bool external = false;            // K supplied by the caller (mfh_matrix_set_upper_triplets): no mesh, scalar variables
    DBuf<int32_t> dElemNodes;
    DBuf<double> dVertPos;

    // ---- material
    int matMode = 0;                  // see k_geometry
    int matKind = MAT_ISO;
    int op = MFH_OP_ELASTICITY;       // operator assembled into K: elasticity (dim x dim blocks) or scalar Laplacian / mass
    std::vector<double> matParams;    // host copy in the layout k_geometry expects
    DBuf<double> dMatParams;
    const double *dMatBorrowed = nullptr;   // linear level of a multigrid hierarchy: the parent's per-element table on the device (the child lives no longer than it)
    DBuf<double> dGeo;
    DBuf<int> dNeg;
    int geoStride = GEO_ISO_STRIDE;
    bool geoValid = false;
    // one assembly pass = embedding kernel + assembly kernel launched back to back and ONE synchronisation at the end (the
    // negative-volume counters and the two kernel times are read then): persistent events and a pinned host word pair
    hipEvent_t passEv[4] = {nullptr, nullptr, nullptr, nullptr};   // embedding start / stop, assembly start / stop
    int *negHost = nullptr;                                        // hipHostMalloc'ed [2]
    bool geoPending = false;                                       // embedding launched, counters not yet checked
    std::vector<double> hGeo;         // lazily downloaded copy (loads / post-processing)
    bool hGeoValid = false;

    // ---- DoF map
    std::vector<int32_t> dofForNode;  // empty = identity
    DBuf<int32_t> dDofForNode;
    bool dofUploaded = false;
    int64_t nDoF = 0;

    // ---- symbolic
    Symbolic sym;
    bool symValid = false;
    bool symHasScatter = false;
    DBuf<int32_t> dRowPtr, dColIdx, dChunkRow, dSpmvChunkRow, dScatter;
    DBuf<int64_t> dContribPtr;
    // Storage of K. upperOnly: only the blocks (r, c >= r) are stored and assembled -- the triangle the reference's TripletMatrix holds
    // (LinearElasticity.hh assembles i <= j; SURVEY.md 8(d) "upper-only variant, matches reference storage"): 55 of an element's 100
    // blocks, half the bytes. Every consumer that needs K through its diagonal blocks, its stored triangle or the Galerkin product works
    // on it; what multiplies by the stored K (the assembled SpMV) needs both triangles.
    // Option "matrix_storage": 1 upper, 0 both, -1 (default) automatic = upper exactly when nothing will multiply by the stored K:
    // elasticity on the matrix-free operator (resolve_upper_storage). upperOnly is the state of the CURRENT symbolic phase.
    int matrixStorage = -1;
    bool upperOnly = false;
    int nCU = 256;                    // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    DBuf<uint32_t> dContribCode;
    DBuf<uint16_t> dContribSlot;
    ShapeTables tables;
    DBuf<double> dPairTable, dMassTable;
    // matrix-free operator (option "matrix_free"): (element, node) pair lists by row chunks
    MfLists mf;
    bool mfValid = false;
    int matrixFree = -1;              // option "matrix_free": 1 on, 0 off, -1 auto = on for elasticity (quadratic: 6x faster than the assembled SpMV; linear, since
                                      // the blocks of the cluster operator hold 512-1024 elements: 0.15 against 0.27 ms at 6.3 M tets, 0.033 against 0.040 at 1 M)
    bool use_mf() const {
        if (external || !haveMesh || hostOnly) return false;
        return matrixFree == 1 || (matrixFree < 0 && op == MFH_OP_ELASTICITY);
    }
    DBuf<int32_t> dMfChunkRow;
    DBuf<int64_t> dMfPairPtr;
    DBuf<uint32_t> dMfPairCode, dMfPairPos;
    DBuf<uint16_t> dMfPairRow;
    MfClusterLists mfc;               // cluster variant (matrix_free_mode 4)
    MfClusterDev mfcDev;
    bool mfcValid = false;
    int mfXcdGroup = 32;              // option "mf_xcd_group": every XCD takes runs of this many consecutive element blocks, all eight inside one
                                      // window (neighbouring blocks share x entries and interface rows in one L2): 0.510 vs 0.527 ms at config 3;
                                      // the same mapping makes the assembly kernel SLOWER (5.42 vs 5.34 ms) and is not used there
    bool mfReorder = true;            // option "mf_reorder": the cluster operator walks the elements in Morton order of their centroids
    DBuf<int32_t> dMfElemPerm, dMfElemNodes;   // perm[new] = old, connectivity in the new order
    int mfLaneStride = 37;            // option "mf_lane_stride": lane t of a block takes element (37 t) % blockElems, so that the lanes of a wave rarely
                                      // add to the same LDS accumulator at once (operator 0.481 vs 0.519 ms at config 3 in one process; 5, 37, 101 alike)
    bool mfGeoFromVerts = true;       // option "mf_geometry_from_vertices": constant materials recompute the gradients in the operator
    int mfBlockElems = 0;             // option "mf_block_elems": elements per block of the cluster variant (0 = 256)
    DBuf<double> dMfSig;              // two-pass operator: per-element nodal forces
    int mfChunkRows = 256, mfChunkPairs = 2048;   // options "mf_chunk_rows" / "mf_chunk_pairs"
    bool mfClusterUnfit = false;      // element order without locality on this mesh: cluster variant not applicable
    int mfModeEff() const { return (mfMode == 4 && mfClusterUnfit) ? 3 : mfMode; }
    int mfMode = 4;                   // option "matrix_free_mode": 4 = cluster variant (default: forces of 256 consecutive elements
                                      // summed in LDS, interface partials only in HBM); 3 = two-pass, forces in list order;
                                      // 2 = two-pass, forces element-major; 1 = per-pair block evaluation (k_spmv_mf)

    // ---- numeric
    DBuf<double> dVals;
    // Option "placement_trials" (0 default): at the first assembly after a symbolic phase the values buffer is allocated up to N more times and
    // the assembly kernel timed on each; the fastest stays. Where the driver puts these bytes moves the kernel between 2.9 and 3.3 ms at 5 M
    // quadratic tets (docs/design/04_2_k_assemble_gather.md (xi)); a caller that assembles hundreds of times may want to pay ~10 ms per trial once.
    int placementTrials = 0;
    int64_t valsGen = 0, placementGen = -1;       // values buffer (re)allocated by the symbolic phase / generation the trials ran for
    std::vector<double> placementMs;              // kernel time on every candidate of the last trials (first = the buffer of the symbolic phase)
    std::string placementNote;        // why the trials stopped early (empty: they did not)
    DBuf<float> dVals32;              // FP32 copy of dVals for the smoother of a multigrid linear level (built by ensure_multigrid, dropped whenever dVals is rewritten)
    bool assembled = false;

    // ---- constraints (SPSDSystem state)
    std::vector<int64_t> fixedVars;
    std::vector<double> fixedVals;
    RawVec<uint8_t> hFixedMask;
    DBuf<uint8_t> dFixedMask;
    DBuf<int64_t> dFixedIdx;
    DBuf<double> dFixedVal;
    bool fixedUploaded = false;
    bool anyFixedNonzero = false;
    bool useGraph = true;             // option "pcg_graph": capture blocks of check_every PCG iterations in a hipGraph
    bool tlSuppress = false;          // solve_one: block-Jacobi for this solve (K singular on the free variables)
    bool refine = true;               // option "refine": iterative refinement when the true residual of a converged solve ends above 2 rtol (solve_one)
    bool solveHomogeneous = false;    // solve_one: treat the fixed values as 0 (columns of the Schur complement)

    // ---- solver
    int precond = MFH_PRECOND_BLOCK_JACOBI;
    // MFH_PRECOND_AUTO (mfh_set_preconditioner): `precond` then holds the choice made for the mesh in hand -- the multigrid V-cycle unless the mesh as
    // a whole is stretched by more than autoStretchMax (option "auto_stretch_max"), where the two-level preconditioner is the faster one (measured
    // crossover at 8 : 1 : 1, docs/design/04_4c_multigrid.md). autoStretch < 0: not looked at yet (new vertices).
    bool precondAuto = false;
    double autoStretch = -1.0, autoStretchMax = 8.0;
    DBuf<double> dDinv;
    bool dinvValid = false;
    DBuf<float> dDinv32;              // FP32 copy of dDinv for the multigrid smoother's fused kernels (smoother_dinv32)
    bool dinv32Valid = false;
    DBuf<double> wx, wr, wz, wp, wAp, wb, wf, wu0, scal, stop;
    // two-level preconditioner (MFH_PRECOND_TWO_LEVEL)
    struct TwoLevel {
        bool valid = false;
        int nModes = 0, nAgg = 0, nColor = 0;
        int64_t m = 0, ldInv = 0;
        double setup_ms = 0, H = 0;
        DBuf<int32_t> aggOfDof, aggPtr, dofsByAgg, colorOfAgg, nbrOfColor, binCoord;
        DBuf<double> relPos, Ainv, rc, yc;
    } tl;
    // p-multigrid preconditioner (MFH_PRECOND_MULTIGRID, mfh_multigrid.cpp): quadratic level -> linear level (a context of its own) ->
    // rigid-body modes of aggregates
    struct AggLevel {                 // an aggregate level: rigid-body modes of lattice bins, operator in lattice-stencil storage
        int64_t nAgg = 0;
        double H = 0, lmax = 0;
        DBuf<int32_t> nbr, parent, coord;     // neighbour table [nAgg][3^dim], parent aggregate on the next level, lattice coordinates [nAgg][3]
        DBuf<int32_t> childPtr, childIdx;     // the inverse of `parent`: the aggregates of every parent, ascending (the transfers towards the parent gather)
        DBuf<double> A, Dinv, rel;            // stencil operator, inverse diagonal blocks, transfer data [nAgg][4] towards the parent
        DBuf<float> A32;                      // A rounded to FP32 for the level's smoother (option mg_coarse_fp32; empty = off)
        DBuf<double> x, b, r, d, t;
        std::vector<int32_t> hCoord;
        std::vector<double> hCentre;
        int nb[3] = {1, 1, 1};               // bins per axis
        int wrap[3] = {0, 0, 0};             // nb of the axes along which the lattice is periodic (periodic DoF maps), else 0
        std::vector<int32_t> hNbr, hParent;  // host copies of nbr / parent (global numbering): the partition of the level is derived from them
        // Row-partitioned contexts: the levels with more than mg_replicate_max aggregates are PARTITIONED (mfh_multigrid.cpp,
        // localize_aggregate_levels): this rank keeps the stencil rows of the aggregates it owns and numbers its aggregates locally --
        // owned ones first (ascending global id), then the halo (aggregates of other ranks that its rows, its DoFs or its children refer
        // to), grouped by owner rank. nbr / parent / childIdx / the vectors are in that numbering; nAgg stays the GLOBAL count.
        bool part = false;
        int64_t nOwn = 0, nLoc = 0;
        std::vector<int32_t> hOwner, hGlobalOf, hLocalOf;   // [nAgg] owner rank (-1: empty bin), [nLoc] global id, [nAgg] local id or -1
        std::vector<int32_t> xPeers;                         // exchange lists of the level (the vocabulary of mfh_dist_setup)
        std::vector<int64_t> xSendPtr{0}, xRecvPtr{0};
        DBuf<int32_t> xSendIdx;
        DBuf<double> xSendBuf, xRecvBuf;
        int64_t rows() const { return part ? nOwn : nAgg; }      // rows this rank smooths
        int64_t size() const { return part ? nLoc : nAgg; }      // entries of its vectors
    };
    struct Multigrid {
        bool valid = false, rigidCoarse = false;
        bool singular = false;                   // built for a K that is singular on the free variables (rigid-motion rows): the dense last level is pinned
        bool linearOnly = false;                 // linear elements: no quadratic level, the context itself is the linear level
        bool distributed = false;                // row-partitioned context: nodal levels partitioned like it (halo exchanges inside the smoothers and the
                                                 // transfers), aggregate levels replicated on every rank (one small all-reduce per application)
        const void *distComm = nullptr;          // the communicator the hierarchy was built on
        DBuf<double> rfull;                      // fine residual with its halo part (restriction reads the edge nodes of halo elements)
        mfh_ctx *coarse = nullptr;    // owned: the linear level
        int64_t nFine = 0, nCoarse = 0;
        DBuf<int32_t> parA, parB, fineOf, resPtr, resIdx;
        DBuf<double> r0, d0, t0, b1, x1, r1, d1, t1;
        int64_t strideAlloc = 0;                 // ... and the spacing the quadratic level's work vectors were sized for (reserve_batch)
        int nrAlloc = 1;                         // right-hand sides the work vectors of every level hold at once (batched V-cycle: they grow on demand)
        double lmax0 = 0, lmax1 = 0, setup_ms = 0;
        // aggregate hierarchy below the linear level (empty: the linear level uses its context's ~1000-aggregate dense coarse space)
        std::vector<std::unique_ptr<AggLevel>> agg;
        DBuf<int32_t> aggOfDof2, aggPtr2, dofsByAgg2;      // linear-level DoFs <-> aggregates of agg[0]
        DBuf<double> relPos2, denseInv;
        int64_t denseM = 0, denseLd = 0;
    } mg;
    int mgSteps0 = 1, mgSteps1 = 1;                  // options "mg_steps_fine" / "mg_steps_coarse": Chebyshev steps before and after the coarse correction
    double mgRatio0 = 0.3, mgRatio1 = 0.3;           // options "mg_ratio_fine" / "mg_ratio_coarse": the smoothers act on [ratio lambda_max, lambda_max]
    int mgCoarseCycles = 1;                          // option "mg_coarse_cycles": cycles of the linear level per application
    double mgEigMargin = 1.1;                        // option "mg_eig_margin": factor on the power-iteration estimates of lambda_max
    int mgAggTarget = 32, mgDenseMax = 1200;         // options "mg_agg_target" (DoFs of the linear level per finest aggregate; 0: no aggregate hierarchy),
                                                     // "mg_dense_max" (aggregates of the level that is inverted densely)
    bool mgCoarseFp32 = true;                        // option "mg_coarse_fp32": inside the multigrid preconditioner the assembled operator of the linear level and the
                                                     // stencil operators of the aggregate levels are READ from FP32 copies (products and sums in FP64): those kernels are
                                                     // bound by the matrix bytes, and a smoother / coarse correction needs no more than single precision of its matrix
    double mgOverCorrection = 1.5;                   // option "mg_over_correction": factor on the corrections prolonged from aggregate levels (piecewise-rigid
                                                     // coarse functions under-estimate smooth corrections; the cycle stays symmetric)
    int mgStepsAgg = 2; double mgRatioAgg = 0.2;     // options "mg_steps_agg" / "mg_ratio_agg": Chebyshev smoother of the aggregate levels
    int mgReplicateMax = 4096;                       // option "mg_replicate_max": row-partitioned contexts partition the aggregate levels with MORE aggregates than this
                                                     // (stencil rows dealt out with the DoFs, one halo exchange per application); smaller levels are replicated on
                                                     // every rank and summed with one small all-reduce. 0: every level replicated (rounds 3-4)
    bool mgAnisotropicBins = false;                  // option "mg_anisotropic_bins": lattice bins of the aggregate levels with the elements' proportions (measured: no gain)
    int mgAggNodes = 0;                              // option "mg_agg_nodes": target DoFs per aggregate of the linear level's coarse space (0 = auto)
    // Chronopoulos-Gear PCG (mfh_solver.cpp): NR interleaved vectors
    DBuf<double> cgU, cgW, cgP, cgS, cgX, cgR, cgF, cgCtl, tlRcN, tlYcN;
    DBuf<double> wNodeField;          // nodal fields on their way to the host (dofToNodeField on the device, solve_multigrid_batch)
    int pcgVariant = -1;              // option "pcg_variant": 1 = Chronopoulos-Gear, 0 = classic PCG, -1 (default) = classic for one
                                      // right-hand side on an unpartitioned context (measured 6-14 % faster per iteration there: one
                                      // vector pass fewer), Chronopoulos-Gear for batches and row-partitioned contexts
    int distPcgVariant = 1;           // option "dist_pcg_variant": 1 = Chronopoulos-Gear (one all-reduce per iteration, default), 0 = classic
                                      // (two all-reduces, one vector pass less per iteration)
    bool mgDinvFp32 = true;           // option "mg_dinv_fp32": the fused kernels of mg_fuse read an FP32 copy of the quadratic level's inverse diagonal blocks (smoother only)
    bool mgFuse = true;               // option "mg_fuse": the PCG loop's residual update / r.z share kernels with the V-cycle's first / last smoothing step (MgFuse)
    bool mgBatch = true;              // option "mg_batch": several right-hand sides under the multigrid preconditioner share the linear and aggregate levels of
                                      // every V-cycle (solve_multigrid_batch); 0: one right-hand side at a time
    bool batchRhs = false;            // option "batch_rhs": several right-hand sides per operator pass (measured slower than one at a time)
    // row-partitioned solve (mfh_dist_setup)
    struct Dist {
        mfh_comm *comm = nullptr;     // not owned
        std::vector<int32_t> peers;
        std::vector<int64_t> sendPtr{0}, recvPtr{0};
        DBuf<int32_t> sendIdx;
        std::vector<int32_t> sendNodesHost;   // host copy of sendIdx (the linear level of the multigrid hierarchy filters it)
        DBuf<double> sendBuf;
        int sendBufW = 0;             // doubles per node the send buffer is sized for
        hipStream_t commStream = nullptr;
        bool commStreamBorrowed = false;      // the stream belongs to the parent context (linear level of a multigrid hierarchy)
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        DBuf<int32_t> opList;         // element blocks / row chunks: interior first, then those reading a halo column
        int64_t nInterior = 0, nBoundary = 0;
        int listKind = 0;             // 0 none, 1 cluster blocks, 2 SpMV chunks
        int64_t listGen = -1;         // mfh_ctx::listsGen the overlap lists were built from
        bool multigridAgreed = false;         // every rank asked for MFH_PRECOND_MULTIGRID (agreed in mfh_dist_solve)
        bool anyFixedNonzeroGlobal = false;   // some rank holds a non-zero fixed value (agreed in mfh_dist_solve)
        int transport = 0;            // what carried the last halo exchange: 1 RCCL send/recv, 2 peer copies, 3 caller callbacks
        int64_t nExchanges = 0;       // halo exchanges since mfh_dist_setup
        // option "dist_profile": timed events around the first operator applications of a solve. ev[0] compute stream before the pack, ev[1] / ev[2]
        // communication stream before / after the exchange, ev[3] compute stream after the interior items, ev[4] after the items that read the halo
        struct Profile { hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; };
        bool profile = false;
        std::vector<Profile> prof = std::vector<Profile>(24);
        int profUsed = 0, profOpen = -1;
        double exchangeMs = 0, interiorMs = 0, boundaryMs = 0, exposedMs = 0, operatorMs = 0;   // averages of the last profiled solve
        int profiled = 0;
    } dist;
    int64_t listsGen = 0;             // bumped whenever the element blocks of the cluster operator or the SpMV row chunks are rebuilt
    int aggNodes = 0;                 // option "agg_nodes": target DoFs per aggregate (0 = auto)
    bool topologyDevice = true;       // option "topology_device": edge numbering + boundary extraction by device radix sorts
    bool symbolicDevice = true;       // option "symbolic_device": build pattern + gather lists on the GPU (element-major order)
    bool tlRapAgg = true;             // option "tl_rap_agg": aggregate-centric Galerkin kernel (0: one wave per row, validation)
    bool tlDeviceAggregates = true;   // option "tl_device_aggregates": build the aggregates on the device (0: host, validation)
    bool tlProbe = false;             // option "tl_probe": build the coarse operator by SpMV probing (validation)
    bool tlHostInverse = false;       // option "tl_host_inverse": invert the coarse operator on the host (validation)
    std::string precondNote;
    int checkEvery = 50;

    // ---- Simulator-level boundary conditions
    std::vector<double> neumannTraction;   // nBE x dim
    RawVec<uint8_t> dirMask;               // nBdryNodes x dim, in the order of mesh.bdryNodes (empty until the first Dirichlet condition)
    RawVec<double> dirVal;                 // nBdryNodes x dim
    std::vector<std::pair<int64_t, std::array<double, 3>>> deltaForces;

    // ---- options
    int asmChunkOrder = 0;           // option "asm_chunk_order": 1 = the assembly visits the row chunks in the order of the elements they gather
                                     // from (an element's contributions then meet in time, its record is fetched once); 0 = in row order
    DBuf<int32_t> dChunkOrder;
    int64_t chunkOrderGen = -1;      // listsGen the order was built from
    bool asmPackedCodes = true;      // option "asm_packed_codes": the device copy of the gather codes is chunk-relative and packed (k_assemble_gather)
    bool codesPacked = false;        // state of dContribCode
    DBuf<int32_t> dChunkElemBase;
    int chunkSlots = 256;            // assembly chunks: 18 KB of LDS accumulators, 8 workgroups per CU
    int contribOrder = 1;            // element-major: 12% faster than rank-major on MI355X (profiles/r01_assembly_variants.md)
    mfh_timing timing{0, 0, 0, 0};

    int dim() const { return mesh.dim; }
    int bs() const { return (op == MFH_OP_ELASTICITY && !external) ? mesh.dim : 1; }   // variables per DoF = block edge of K
    int asmMat() const { return op == MFH_OP_ELASTICITY ? matKind : (op == MFH_OP_LAPLACIAN ? (int)MAT_LAPLACE : (int)MAT_MASS); }
    int64_t nOwnedDoFSet = -1;        // mfh_dof_map_partitioned: the first nOwnedDoFSet DoFs are this rank's rows (-1: not set)
    int64_t nOwnedDoF() const {
        // owned rows: DoFs of the first nOwned nodes; with a DoF map all DoFs are owned unless mfh_dof_map_partitioned said otherwise
        if (nOwnedDoFSet >= 0) return nOwnedDoFSet;
        if (mesh.nOwned == mesh.nNode) return nDoF;
        return mesh.nOwned;
    }
};

namespace mfhi {   // internal helpers with external linkage (defined in mfh_api.cpp unless inline)
using namespace mfh;

#define MFH_TRY(ctx) try {         \
    mfh::k::DetScope detScope_(mfhi::det_buf(ctx)); \
    mfh::PoolScope poolScope_((ctx) ? mfhi::ctx_stream(ctx) : nullptr, (ctx) ? mfhi::ctx_comm_stream(ctx) : nullptr, ((ctx) && !mfhi::ctx_host_only(ctx)) ? 1 : 0);
#define MFH_CATCH(ctx)                                              \
    } catch (const mfh::Error &e) {                                 \
        if (ctx) (ctx)->err = e.what();                             \
        return e.code;                                              \
    } catch (const std::bad_alloc &) {                              \
        if (ctx) (ctx)->err = "host allocation failed";             \
        return MFH_ERR_HIP;                                         \
    } catch (const std::exception &e) {                             \
        if (ctx) (ctx)->err = e.what();                             \
        return MFH_ERR_INVALID;                                     \
    }                                                               \
    return MFH_OK;

// the scratch of the reproducible global sums for the launches of this API call (null members unless option "deterministic" is on)
inline mfh::k::DetBuf det_buf(const mfh_ctx *c) {
    mfh::k::DetBuf d;
    if (c && c->deterministic && c->detPartials.p) { d.partials = c->detPartials.p; d.counter = c->detCounter.p; d.cap = (int)((c->detPartials.n - 8) / 4); }
    return d;
}

inline hipStream_t ctx_stream(const mfh_ctx *c) { return c->stream; }
inline hipStream_t ctx_comm_stream(const mfh_ctx *c) { return c->dist.commStream; }
inline bool ctx_host_only(const mfh_ctx *c) { return c->hostOnly; }

inline void require(bool cond, mfh_status code, const char *msg) {
    if (!cond) throw Error(code, msg);
}
inline void require_device(const mfh_ctx *c);

struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s;
    explicit EventTimer(hipStream_t s_) : s(s_) {
        MFH_HIP(hipEventCreate(&a));
        MFH_HIP(hipEventCreate(&b));
        MFH_HIP(hipEventRecord(a, s));
    }
    double stop() {
        MFH_HIP(hipEventRecord(b, s));
        MFH_HIP(hipEventSynchronize(b));
        float ms = 0;
        MFH_HIP(hipEventElapsedTime(&ms, a, b));
        return ms;
    }
    ~EventTimer() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
};

inline void require_device(const mfh_ctx *c) {
    if (c->hostOnly)
        throw Error(MFH_ERR_HIP, "host-only context (device -1): no HIP device, and there is no CPU fallback");
}

// The recurrence residual of CG can drift away from the true one (a singular system with an inconsistent right-hand side: the
// iterate grows along the null space, the recursively updated r keeps shrinking): "converged" then means nothing. CHOLMOD
// would have refused such a matrix; report it instead of returning garbage.
inline void check_residual_gap(const mfh_solve_info &li, double rtol) {
    if (li.converged && li.true_rel_residual > std::max(1000.0 * rtol, 1e-4))
        throw Error(MFH_ERR_NOT_CONVERGED, "PCG breakdown (residual gap): the recurrence residual reached " + std::to_string(li.rel_residual) +
                                               " but the true residual ||f - K u|| / ||f|| is " + std::to_string(li.true_rel_residual) +
                                               ": K is singular on the free variables with an inconsistent right-hand side, or too ill-conditioned");
}

// the storage the next symbolic phase will build (see mfh_ctx::matrixStorage)
inline bool resolve_upper_storage(const mfh_ctx *c) {
    if (c->matrixStorage >= 0) return c->matrixStorage == 1;
    return c->haveMesh && !c->hostOnly && !c->external && c->op == MFH_OP_ELASTICITY && c->use_mf() && !c->tlProbe && c->tlRapAgg;
}

// consumers that multiply by the stored K cannot work on the upper-only storage (option "matrix_storage" 1; the automatic
// choice never puts them there)
inline void require_full_storage(const mfh_ctx *c, const char *what) {
    if (c->upperOnly) throw Error(MFH_ERR_UNSUPPORTED, std::string(what) + " needs both triangles of K: set option matrix_storage to 0");
}

inline int32_t dof_of(const mfh_ctx *c, int64_t node) { return c->dofForNode.empty() ? (int32_t)node : c->dofForNode[node]; }
void invalidate_matrix(mfh_ctx *c);
void ensure_dirichlet_tables(mfh_ctx *c);
void dist_detach(mfh_ctx *c);
bool dist_active(const mfh_ctx *c);
int dist_rank(const mfh_ctx *c);
int dist_world(const mfh_ctx *c);
void dist_apply(mfh_ctx *c, double *x, double *y, bool masked);
void dist_halo(mfh_ctx *c, double *v, int W);
void dist_allreduce(mfh_ctx *c, double *dev, int64_t n);
void dist_level_forward(mfh_ctx *c, mfh_ctx::AggLevel &A, double *v, int W);        // owned entries -> the halo copies on the other ranks
void dist_level_reverse_add(mfh_ctx *c, mfh_ctx::AggLevel &A, double *v, int W);    // halo partial sums -> their owners, added there
void dist_exchange_lists(mfh_ctx *c, const std::vector<std::vector<int32_t>> &toRank, std::vector<std::vector<int32_t>> &fromRank);   // setup: variable-length lists, all to all
void dist_setup_child(mfh_ctx *c, mfh_ctx *child, const std::vector<int32_t> &keep);
void dist_agree(mfh_ctx *c);
void dist_profile_collect(mfh_ctx *c);
void refresh_storage_rule(mfh_ctx *c);
void reset_bcs(mfh_ctx *c);
void clear_fixed(mfh_ctx *c);
void add_fixed(mfh_ctx *c, int64_t n, const int64_t *vars, const double *vals);
void ensure_geometry(mfh_ctx *c, bool deferCheck = false);
void finish_geometry(mfh_ctx *c);
void ensure_mf_cluster(mfh_ctx *c);
void ensure_precond(mfh_ctx *c);
bool ensure_twolevel(mfh_ctx *c);
bool ensure_multigrid(mfh_ctx *c);
void ensure_coarse_levels(mfh_ctx *c, int nrhs);
void destroy_multigrid(mfh_ctx *c);
// What the PCG loop around the V-cycle takes over from it / hands to it when the first level's smoother is ONE step (mg_fuse_scale > 0):
//   presmoothed: z already holds zs Dinv r (k_pcg_update's ZS flavour wrote it with the residual update) -- the cycle starts at its residual;
//   rzScal:      the cycle's last kernel also forms r.z into the iteration's history (k_mg_cheb_rz; rzMask: the fixed-variable mask, may be null)
//                -- the loop launches no k_mg_rz. Loop k of a batch: history rzScal + k scalStride.
struct MgFuse { bool presmoothed = false; double *rzScal = nullptr; const uint8_t *rzMask = nullptr; };
const float *smoother_dinv32(mfh_ctx *c);     // the FP32 copy of the inverse diagonal blocks (made on first use after every change of dDinv), or null with option mg_dinv_fp32 0
double mg_fuse_scale(const mfh_ctx *c);       // zs = 1 / theta of the first level's smoother when the fusion applies (unpartitioned quadratic hierarchy, one step), else 0
void mg_precond(mfh_ctx *c, const double *r, double *z, const double *scal, int it, const double *stop, const MgFuse *fuse = nullptr);
void mg_precond_batch(mfh_ctx *c, int NR, const double *r, double *z, int64_t vecStride, const double *scal, int64_t scalStride, int it, const double *stop,
                      const MgFuse *fuse = nullptr);
k::AsmArgs asm_args(mfh_ctx *c);
const int32_t *device_dof_map(mfh_ctx *c);
void box_corners(mfh_ctx *c, const double *mn, const double *mx, int relative, double *omn, double *omx);
void dirichlet_vars(mfh_ctx *c, std::vector<int64_t> &vars, std::vector<double> &vals);
int64_t pin_node(const mfh_ctx *c);
void solve_one(mfh_ctx *c, const double *f, double *u, double rtol, int maxit, mfh_solve_info *info);
void solve_many(mfh_ctx *c, int nrhs, const double *f, double *u, int64_t stride, double rtol, int maxit, mfh_solve_info *infos);
// Device-resident ends of a multigrid batch (mfh_solve_cell_problems): the right-hand sides are constantStrainLoad vectors formed on the device
// (no host load vector exists), the solutions leave as nodal fields (dofToNodeField on the device, one download per field)
struct BatchIO {
    const double *cstrains = nullptr;     // NR x flatLen constant strains (flattened, tensor shear); null: right-hand sides come from the host
    double *uNodes = nullptr;             // NR nodal fields, nodeStride doubles apart; null: DoF vectors to u
    int64_t nodeStride = 0;
};
bool device_rhs_supported(mfh_ctx *c);            // solve_one(c, nullptr, ...) -- right-hand side already in c->wf -- will work for this context
bool multigrid_batch_ready(mfh_ctx *c, int nrhs);     // the hierarchy exists (built if need be) and the batched V-cycle applies to this context
void solve_multigrid_batch(mfh_ctx *c, int NR, const double *f, double *u, int64_t stride, double rtol, int maxit, mfh_solve_info *infos, const BatchIO *io = nullptr);
bool constant_strain_load_device(mfh_ctx *c, const double *cstrainFlat, double *outDev);   // through the cluster operator's lists; false: not available
}   // namespace mfhi
