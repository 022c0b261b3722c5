// Internal state of a libmeshfem_hip context. Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>
#include <chrono>
#include <thread>
#include <functional>
#include <algorithm>
#include <array>
#include <memory>
#include <cstdlib>
#include "../../include/meshfem_hip.h"
#include "../../include/meshfem_hip_extras.h"

namespace mfh {

struct Error : std::runtime_error {
    mfh_status code;
    Error(mfh_status c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define MFH_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            throw mfh::Error(MFH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

inline double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// Named ranges for rocprofv3 --marker-trace under the reference's timer-section names (GlobalBenchmark.hh:14-34:
// "Assemble System", "Compress Matrix", "Elasticity Solve"). libroctx is looked up with dlopen the first time a range is
// opened and only if MFH_ROCTX=1, so the library has no link-time dependency on the profiler.
struct RoctxRange {
    explicit RoctxRange(const char *name);
    ~RoctxRange();
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
private:
    bool active = false;
};

// Simple parallel-for over [0, n) in contiguous ranges (host setup code only).
void parallel_ranges(int64_t n, const std::function<void(int64_t, int64_t, int)> &f, int64_t minGrain = 4096);
// madvise(MADV_HUGEPAGE) on the 2 MiB-aligned interior of a fresh host range (no-op below 8 MiB or with MFH_HOST_HUGE_PAGES=0): the tables of a
// 40 M-element mesh are ~4 GB of untouched memory, i.e. a million 4 KiB page faults without it (mfh_mesh.cpp)
void host_advise_huge_pages(void *p, size_t bytes);
int  host_threads();

// device memory of the library: an arena that splits and coalesces inside the hipMalloc segments it holds (mfh_pool.cpp)
void *device_alloc(size_t bytes);
void device_free(void *p);
void device_cache_trim();
void device_arena_context_opened(int dev);     // mfh_create / mfh_destroy of a device context: when a context closes the arena is trimmed
void device_arena_context_closed(int dev);     // to its live high-water mark, when the last one closes to a small reserve
// one free segment of that size from the driver, now (on a thread of its own if async); cls 2: a segment for the value array of K (PoolTag);
// false: a synchronous reservation did not get its memory
bool device_arena_reserve(int dev, size_t bytes, bool async, int cls = 1);
// Requests made while a PoolTag(1) is alive are the value array of K: served from segments that hold nothing else (mfh_pool.cpp, class 2)
struct PoolTag {
    int saved;
    explicit PoolTag(int tag);
    ~PoolTag();
    PoolTag(const PoolTag &) = delete;
    PoolTag &operator=(const PoolTag &) = delete;
};
size_t device_arena_largest_free(int dev);
void device_arena_stats(int dev, int64_t out[8]);   // held, live, live high-water mark, segments, free chunks, bytes returned to the driver, quarantined, bound
// What a release waits for before the block may be handed to somebody else: the streams of the context the calling thread is working for
// (PoolScope, installed by every API entry), not the whole device -- hipDeviceSynchronize from one host thread invalidates a stream
// capture another thread has open (tests/test_gpu_threads.py). Outside any scope: the device.
struct PoolScope {
    hipStream_t saved[2];
    int savedMode;
    PoolScope(hipStream_t a, hipStream_t b, int mode);     // mode 1: synchronise a and b; 2: nothing (the caller has synchronised)
    ~PoolScope();
    PoolScope(const PoolScope &) = delete;
    PoolScope &operator=(const PoolScope &) = delete;
};
void device_cache_stats(int dev, int64_t *cachedBytes, int64_t *blocks, int64_t *hits, int64_t *misses, int64_t *flushes);

// Device buffer with explicit size tracking.
// K's value array: tiles of 64 consecutive slots x NB components (a wave reads / writes 512 contiguous bytes per component). MFH_TILE_LOG / MFH_TILE_PAD
// (doubles of padding behind every tile) exist for the tile-geometry probe of round 6 (timing-only builds; docs/design/04_2 (xii)): the default build has 6 / 0.
#ifndef MFH_TILE_LOG
#define MFH_TILE_LOG 6
#endif
#ifndef MFH_TILE_PAD
#define MFH_TILE_PAD 0
#endif
inline size_t tiled_count(int64_t slots, int NB) {      // doubles of a tiled array of `slots` slots
    const size_t T = (size_t)1 << MFH_TILE_LOG;
    return (((size_t)slots + T - 1) >> MFH_TILE_LOG) * (T * (size_t)NB + MFH_TILE_PAD);
}

template <class T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
    DBuf() = default;
    DBuf(const DBuf &) = delete;
    DBuf &operator=(const DBuf &) = delete;
    ~DBuf() { release(); }
    void release() {
        if (p) device_free(p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        if (count == n && p) return;
        release();
        if (count == 0) return;
        p = (T *)device_alloc(count * sizeof(T));
        n = count;
    }
    void swap(DBuf &o) { std::swap(p, o.p); std::swap(n, o.n); }
    // grow-only: work vectors whose size alternates between calls (batch widths 2, 1, 2, ...) are not reallocated every time
    void reserve(size_t count) {
        if (count > n || !p) alloc(count);
    }
    void upload(const T *h, size_t count, hipStream_t s) {
        alloc(count);
        if (count) {
            MFH_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
            MFH_HIP(hipStreamSynchronize(s));
        }
    }
    template <class A> void upload(const std::vector<T, A> &h, hipStream_t s) { upload(h.data(), h.size(), s); }
    void download(T *h, size_t count, hipStream_t s) const {
        if (count) {
            MFH_HIP(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
            MFH_HIP(hipStreamSynchronize(s));
        }
    }
    void zero(hipStream_t s) {
        if (n) MFH_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s));
    }
};

// ------------------------------------------------------------------------------------------------
// Element-type constants
// ------------------------------------------------------------------------------------------------
inline int nodes_per_elem(int dim, int deg) {
    if (dim == 3) return deg == 1 ? 4 : 10;
    return deg == 1 ? 3 : 6;
}
inline int nodes_per_bdry_elem(int dim, int deg) {
    if (dim == 3) return deg == 1 ? 3 : 6;
    return deg == 1 ? 2 : 3;
}
inline int flat_len(int dim) { return dim * (dim + 1) / 2; }
// Simplex.hh:43-44
static const int kEdgeStart[6] = {0, 1, 2, 0, 2, 1};
static const int kEdgeEnd[6] = {1, 2, 0, 3, 3, 3};

// geometry record per element (doubles): gradLambda (dim*(dim+1), vertex-major: gl[k*dim+a] =
// d lambda_k / d x_a), volume, then material: iso -> lambda, mu ; general -> 21 (3D) / 6 (2D)
// upper-triangular entries of D row-major. Stride padded to a multiple of 2 doubles.
constexpr int GEO_ISO_STRIDE = 16;      // 12 + 1 + 2 (+1 pad)  = 128 B, one cache line
constexpr int GEO_GEN_STRIDE = 36;      // 12 + 1 + 21 (+2 pad) = 288 B
constexpr int GEO_ORTHO_STRIDE = 24;    // 12 + 1 + 6 normal + 3 shear (+2 pad) = 192 B (orthotropic field: the zeros of D are not stored).
                                        // A 256-B aligned record (stride 32) was measured at config 4: assembly 2.39 vs 2.44 ms, but the
                                        // matrix-free operator, which runs every PCG iteration, 0.310 vs 0.291 ms -- 192 B it is.

// MAT_LAPLACE / MAT_MASS: scalar operators on the same machinery (1x1 blocks; Laplacian.hh:27-57, MassMatrix.hh:50-86)
enum MaterialKind { MAT_ISO = 0, MAT_GENERAL = 1, MAT_LAPLACE = 2, MAT_MASS = 3, MAT_ORTHO = 4 };
inline bool mat_is_scalar(int mat) { return mat == MAT_LAPLACE || mat == MAT_MASS; }

// ------------------------------------------------------------------------------------------------
// Host-side mesh (FEMMesh restatement)
// ------------------------------------------------------------------------------------------------
// The big per-element / per-node tables are filled by parallel loops right after they are sized: a vector whose resize() leaves
// the new entries uninitialised saves the single-threaded zero fill (25 + 20 ms of the FEMMesh build at 5 M quadratic tets).
template <class T>
struct DefaultInitAllocator : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAllocator<U>; };
    using std::allocator<T>::allocator;
    DefaultInitAllocator() = default;
    template <class U> DefaultInitAllocator(const DefaultInitAllocator<U> &) {}
    template <class U> void construct(U *ptr) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(ptr)) U; }
    template <class U, class... Args> void construct(U *ptr, Args &&...args) { ::new (static_cast<void *>(ptr)) U(std::forward<Args>(args)...); }
};
template <class T> using RawVec = std::vector<T, DefaultInitAllocator<T>>;
// size a RawVec and take its first-touch page faults on all host threads (one write per page): a download or a single-threaded fill that
// follows then runs at memory speed instead of at the page-fault rate of one thread (0.97 GB of edge-node ranks at 119^3: ~0.2 s)
template <class T> void resize_prefaulted(RawVec<T> &v, size_t n) {
    const size_t old = v.size();
    v.resize(n);
    if (n <= old) return;
    char *base = reinterpret_cast<char *>(v.data());
    const int64_t b0 = (int64_t)(old * sizeof(T)), b1 = (int64_t)(n * sizeof(T));
    host_advise_huge_pages(base + b0, (size_t)(b1 - b0));
    const int64_t pages = (b1 - b0 + 4095) / 4096;
    parallel_ranges(pages, [&](int64_t pb, int64_t pe, int) {
        for (int64_t q = pb; q < pe; ++q) base[std::min<int64_t>(b0 + q * 4096, b1 - 1)] = 0;
    }, 1024);
}

struct HostMesh {
    int dim = 0, deg = 0, npe = 0, npbe = 0;
    int64_t nElem = 0, nNode = 0, nVert = 0, nOwned = 0;
    RawVec<int32_t> elemNodes;          // nElem x npe
    RawVec<double> vertPos;             // nVert x dim
    RawVec<double> nodePos;             // nNode x dim (built lazily for mesh_set)
    bool hasTopology = false;
    // boundary (only with topology)
    std::vector<int32_t> bdryElemNodes; // nBE x npbe (volume node ids)
    std::vector<int32_t> bdryParent;    // nBE: the volume element each boundary element is a face / edge of
    std::vector<int32_t> bdryNodes;     // volume node ids in boundary-node order
    std::vector<uint8_t> isBdryNode;    // nNode
    std::vector<double> bdryVol;        // nBE
    std::vector<double> bdryNormal;     // nBE x dim
    std::vector<uint8_t> bdryInternal;  // nBE (periodic)
    int64_t nBE() const { return npbe ? (int64_t)bdryElemNodes.size() / npbe : 0; }
};

// dElemNodesOut / dNodePosOut (device topology only): the device copies of the node table and the node positions, built on the device
// from the vertices; *deviceTables says whether they were written
void build_fem_mesh(HostMesh &m, int dim, int deg, int64_t nElem, int64_t nVert, const int32_t *elemVerts,
                    const double *vertPos, bool useDevice = false, hipStream_t stream = nullptr, DBuf<int32_t> *dElemNodesOut = nullptr,
                    DBuf<double> *dNodePosOut = nullptr, bool *deviceTables = nullptr);
// device topology (mfh_symbolic_gpu.hip); false = mesh does not fit the packed sort keys, use the host path
bool build_topology_device(int dim, int deg, int64_t nElem, int64_t nVert, const int32_t *hostElemVerts, hipStream_t s,
                           RawVec<int32_t> &instEdge, int32_t &nEdgeNodes, std::vector<uint32_t> &bdryInst,
                           const double *hostVertPos = nullptr, DBuf<int32_t> *dElemNodesOut = nullptr, DBuf<double> *dNodePosOut = nullptr,
                           // host work to run while the device is busy: (0, 0) once the edge sort is queued, (1, nEdgeNodes) once the half-face sort is
                           const std::function<void(int, int32_t)> &hostOverlap = {});
void compute_node_positions(HostMesh &m);
void compute_boundary_geometry(HostMesh &m, const double *vertPos);
void periodic_dof_map(const HostMesh &m, double eps, std::vector<int32_t> &dofForNode, int64_t &nDoF,
                      std::vector<uint8_t> &bdryInternal, bool ignoreMismatch = false, int ignoreDimsMask = 0);

// ------------------------------------------------------------------------------------------------
// Symbolic structure (BSR pattern + gather lists)
// ------------------------------------------------------------------------------------------------
struct Symbolic {
    int64_t nRows = 0;                  // block rows (owned DoFs)
    int64_t nCols = 0;                  // block cols (all local DoFs)
    int64_t nnzb = 0;                   // STORED blocks
    int64_t nMirror = 0;                // stored blocks (r, c) with r < c < nRows: with upper-only storage the logical K has nnzb + nMirror blocks
    RawVec<int32_t> rowPtr;             // nRows+1 (RawVec: sized, pre-faulted by the host threads, then downloaded)
    std::vector<int32_t> colIdx;        // nnzb
    // row chunks: consecutive rows whose slot count <= chunkSlots
    std::vector<int32_t> chunkRow;      // nChunk+1   (assembly chunks)
    std::vector<int32_t> spmvChunkRow;  // row chunks of the SpMV kernel (<= spmvChunkSlots blocks)
    int spmvChunkSlots = 0;
    // gather lists, grouped by chunk
    std::vector<int64_t> contribPtr;    // nChunk+1
    std::vector<uint32_t> contribCode;  // e*npe*npe + i*npe + j
    std::vector<uint16_t> contribSlot;  // slot - rowPtr[chunkRow[c]]
    // element-major scatter map (built on demand for the atomic variant)
    std::vector<int32_t> scatterSlot;   // nElem x npe x npe, -1 where row not owned
    int chunkSlots = 0;
    int maxRowLen = 0;
    int64_t nChunk() const { return (int64_t)chunkRow.size() - 1; }
};

void build_symbolic(const HostMesh &m, const std::vector<int32_t> &dofForNode /* empty = identity */,
                    int64_t nDoF, int64_t nOwnedDoF, int chunkSlots, int contribOrder, bool wantScatter,
                    Symbolic &S, bool upperOnly = false);

// device implementation (mfh_symbolic_gpu.hip): element-major gather lists via two radix sorts
// gather lists of the matrix-free operator (mfh_symbolic_gpu.hip)
struct MfLists {
    std::vector<int32_t> chunkRow;
    std::vector<int64_t> pairPtr;
    int maxRows = 0;
    int64_t nPairs = 0;
};
// Cluster variant of the matrix-free operator: elements in blocks of MF_BLOCK consecutive elements; the nodal forces of
// a block are summed in LDS. Rows whose elements all lie in one block are finished there; the others (interface rows)
// leave one partial sum per (block, row) in a small buffer that a second pass sums.
constexpr int MF_BLOCK = 256;             // threads of a workgroup of the cluster operator
constexpr int MF_BLOCK_ELEMS_MAX = 4096;  // elements of a block: the lanes take them in rounds of MF_BLOCK (default 512 for quadratic elements, more for linear ones)
struct MfClusterLists {
    int64_t nBlocks = 0, nEntries = 0, nIface = 0, nIfaceRows = 0;
    int64_t nChunk = 0;                 // chunks of the second pass (chunkRow / pairPtr live on the device only: MfClusterDev)
    int blockElems = MF_BLOCK;          // elements per block (<= MF_BLOCK_ELEMS_MAX)
    int maxLocal = 0;                   // largest number of distinct rows of a block (LDS accumulators)
    int maxRows = 0;                    // most interface rows of a second-pass chunk (LDS of k_mf_rows)
};
struct MfClusterDev {
    DBuf<uint16_t> localIdx;            // [nElem*npe]: index of the pair's row among the block's rows
    DBuf<int32_t> blockPtr;             // [nBlocks+1] into the entry arrays
    DBuf<int32_t> entryRow;             // [nEntries] global row of the entry
    DBuf<int32_t> entryDest;            // [nEntries] -1: finish here (write y), >= 0: slot in the interface buffer, -2: row not owned
    DBuf<uint16_t> ifaceRow;            // [nIface] row - chunkRow[chunk] of every interface slot (row order)
    DBuf<uint8_t> rowIsIface;           // [nRows]
    DBuf<int32_t> rowMap;               // [nIfaceRows] global row of every interface row (the second pass works in this numbering)
    DBuf<int32_t> chunkRow;
    DBuf<int64_t> pairPtr;
    DBuf<double> ifaceBuf;              // [nIface * dim]
    DBuf<int32_t> elemPtr;              // [nBlocks+1] variable-size blocks (whole cells), empty = uniform blocks
};
// blockStart (may be null): element offsets of variable-size blocks [nBlocks+1], every block <= MF_BLOCK elements
void build_mf_cluster_lists_device(const HostMesh &m, const int32_t *dElemNodes, const int32_t *dDofForNode, int64_t nRows, hipStream_t s,
                                   MfClusterLists &L, MfClusterDev &D, int blockElems = MF_BLOCK, const std::vector<int32_t> *blockStart = nullptr);
void build_element_order_device(const HostMesh &m, const int32_t *dElemNodes, const double *dNodePos, hipStream_t s, DBuf<int32_t> &perm,
                                DBuf<int32_t> &elemNodesOut, int maxBlock, std::vector<int32_t> &blockStart);
void build_mf_lists_device(const HostMesh &m, const int32_t *dElemNodes, const int32_t *dDofForNode, int64_t nRows, hipStream_t s,
                           MfLists &L, DBuf<uint32_t> &dPairCode, DBuf<uint16_t> &dPairRow, DBuf<uint32_t> &dPairPos, int maxRowsCap = 256,
                           int maxPairs = 2048);
// greedy row chunks of at most chunkSlots slots (whole rows); breaks = rows (ascending) at which a chunk must end; scanned by the host threads in
// ranges of `grain` rows and stitched to the sequential result (mfh_symbolic_gpu.hip)
std::vector<int32_t> make_chunks(const int32_t *rowPtr /* nRows + 1 */, int64_t nRows, int chunkSlots, const std::vector<int64_t> &breaks = {},
                                 int64_t grain = (int64_t)1 << 18, int maxThreads = 0);
void build_symbolic_device(const HostMesh &m, const int32_t *dElemNodes, const int32_t *dDofForNode, int64_t nDoF, int64_t nOwnedDoF,
                           int chunkSlots, bool wantScatter, hipStream_t s, Symbolic &S, DBuf<int32_t> &dRowPtr, DBuf<int32_t> &dColIdx,
                           DBuf<uint32_t> &dContribCode, DBuf<uint16_t> &dContribSlot, DBuf<int32_t> &dScatter, bool upperOnly = false,
                           DBuf<int32_t> *dChunkElemBase = nullptr, bool *codesPacked = nullptr);

// ------------------------------------------------------------------------------------------------
// P2 coefficient tables: for node i, grad phi_i(q) = alpha_i(q) gl[s_i] + beta_i(q) gl[t_i]
// (EmbeddedElement.hh:315-332). pairTable[(i*npe+j)*4 + {aa,ab,ba,bb}] = sum_q w_q coef_i coef_j
// with the reference's quadrature rule of degree 2(deg-1) (GaussQuadrature.hh:115-127,283-295).
// ------------------------------------------------------------------------------------------------
struct ShapeTables {
    int npe = 0;
    int sup_s[10], sup_t[10];
    std::vector<double> pairTable;      // npe*npe*4
    double pairConst[6] = {1, 0, 0, 0, 0, 0}; // [2*type + (a != b)], type 0/1/2 = two/one/no vertex-offset terms
    std::vector<double> intGrad;        // npe*2: integral (unit volume) of alpha_i, beta_i  (constantStrainLoad)
    std::vector<double> massRef;        // npe*npe: integral of phi_i phi_j over the unit-volume simplex (exact)
};
void build_shape_tables(int dim, int deg, ShapeTables &T);

// ------------------------------------------------------------------------------------------------
// Two-level preconditioner (mfh_twolevel.cpp)
// ------------------------------------------------------------------------------------------------
struct Aggregates {
    int dim = 0, nAgg = 0, nColor = 0;
    int nb[3] = {1, 1, 1};               // bins of the lattice per axis
    double H = 0;                        // bin edge (rotation modes are scaled by 1/H)
    bool binsTooFew = false;
    std::vector<int32_t> aggOfDof;       // nDoF
    std::vector<double> centroid;        // nAgg x 3
    std::vector<int32_t> aggPtr;         // nAgg+1
    std::vector<int32_t> dofsByAgg;      // nDoF, grouped by aggregate
    std::vector<int32_t> colorOfAgg;     // nAgg
    std::vector<int32_t> binCoord;       // nAgg x 3 lattice coordinates
    std::vector<int32_t> nbrOfColor;     // nAgg x nColor: the neighbour (or self) of that colour, -1 if none
};
void build_aggregates(int dim, int64_t nDoF, const std::vector<double> &dofPos, int targetNodes, Aggregates &A);
// lattice tables (colours, bin coordinates, neighbours by colour) from the compact bin numbering
void aggregate_lattice_tables(int dim, const int nb[3], const std::vector<int32_t> &binId, Aggregates &A);
// the same aggregates built on the device (mfh_symbolic_gpu.hip): bins, DoFs by aggregate (radix sort), centroids and the
// relative positions stay in HBM; only the bin occupancy and the per-aggregate counts visit the host
void dof_positions_device(int64_t nNode, int dim, const int32_t *dDofForNode, const double *dNodePos, int64_t nDoF, hipStream_t s,
                          DBuf<double> &out);
void build_aggregates_device(int dim, int64_t nDoF, const double *dPos, int targetNodes, hipStream_t s, Aggregates &A,
                             DBuf<int32_t> &dAggOfDof, DBuf<double> &dRelPos, DBuf<int32_t> &dAggPtr, DBuf<int32_t> &dDofsByAgg,
                             const double *globalBox = nullptr, int64_t globalCount = 0, bool fullLattice = false, const double *aspect = nullptr);
void element_extent_sums_device(int dim, int64_t nElem, int npe, const int32_t *dElemNodes, const double *dPos, hipStream_t s, double out[3]);
void bounding_box_device(int dim, int64_t nDoF, const double *dPos, hipStream_t s, double mn[3], double mx[3]);
void wrap_positions_device(int64_t n, int dim, const double box[6], hipStream_t s, double *dPos, int skipDims = 0);
bool spd_inverse_inplace(int64_t n, double *A);
// transfer lists of the p-multigrid preconditioner on the device (meshes in the library's own numbering, identity DoF map)
void build_mg_transfer_device(const HostMesh &m, const int32_t *dElemNodes, hipStream_t s, DBuf<int32_t> &parA, DBuf<int32_t> &parB,
                              DBuf<int32_t> &fineOf, DBuf<int32_t> &resPtr, DBuf<int32_t> &resIdx);

} // namespace mfh

// ------------------------------------------------------------------------------------------------
// Kernel launchers (mfh_kernels.hip: element kernels and operators; mfh_kernels_solver.hip: preconditioners, dense inverse, PCG vectors)
// ------------------------------------------------------------------------------------------------
namespace mfh { namespace k {

// option "deterministic": scratch of the run-to-run reproducible global sums (commit_sums, mfh_device.hh). The launchers read the
// calling thread's current buffer (a context's solve installs its own for the duration: DetScope), so that no launch signature changes.
struct DetBuf {
    double *partials = nullptr;      // [cap][4] workgroup partials
    unsigned *counter = nullptr;     // arrival ticket
    int cap = 0;                     // workgroups the scratch holds: deterministic launches use at most this many
};
extern thread_local DetBuf t_det;
struct DetScope {
    DetBuf saved;
    explicit DetScope(const DetBuf &d) : saved(t_det) { t_det = d; }
    ~DetScope() { t_det = saved; }
    DetScope(const DetScope &) = delete;
    DetScope &operator=(const DetScope &) = delete;
};
inline int det_grid(int grid) { return t_det.partials ? std::min(grid, t_det.cap) : grid; }
// the second stage of a deterministic global sum: call right after a launch whose kernel ends in commit_sums (no-op unless t_det is set)
void launch_det_finish(hipStream_t s);

struct AsmArgs {
    int dim, deg, npe, mat;             // mat: MaterialKind
    int geoStride;
    const double *geo;                  // nElem x geoStride
    const double *pairTable;            // device copy of ShapeTables::pairTable
    const double *massTable;            // device copy of ShapeTables::massRef (npe x npe, unit-volume element)
    double pairConst[6];                // the distinct pair coefficients (ShapeTables::pairConst)
    // gather
    int64_t nChunk;
    const int32_t *chunkRow;
    const int32_t *rowPtr;
    const int64_t *contribPtr;
    const uint32_t *contribCode;
    const uint16_t *contribSlot;
    int chunkSlots;
    // atomic
    int64_t nElem;
    const int32_t *scatterSlot;
    // output
    double *vals;                       // tiled [ceil(nnzb/64)][dim*dim][64]
    int64_t nnzb;
    int xcd;                            // 1: XCD-contiguous chunk mapping (xcd_item)
    const int32_t *chunkElemBase;       // packed gather codes: (element - chunkElemBase[chunk]) << 7 | ij (nullptr: absolute codes e npe^2 + ij)
    const int32_t *chunkOrder;          // workgroup b takes chunk chunkOrder[b] (nullptr: b): chunks visited in element order
    int upperOnly;                      // the lists cover the blocks (r, c >= r) only (names the kernel instantiation)
    int det;                            // option "deterministic": the waves of a workgroup add their contributions in wave order
};

void launch_geometry(int dim, int deg, int mat, int64_t nElem, const int32_t *elemNodes, int npe,
                     const double *vertPos, const double *matParams, int matMode, double *geo, int geoStride,
                     int *negCount, hipStream_t s);
void launch_assemble_gather(const AsmArgs &a, hipStream_t s);
void launch_chunk_keys(const AsmArgs &a, uint32_t *keys, hipStream_t s);
bool launch_pack_codes(int64_t nChunk, const int64_t *contribPtr, uint32_t *contribCode, int npe, int32_t *chunkElemBase, int *flag, hipStream_t s);
void launch_assemble_atomic(const AsmArgs &a, hipStream_t s);
void launch_element_stiffness(const AsmArgs &a, int64_t first, int64_t count, double *KeOut, hipStream_t s);
// deltaP != nullptr: the discrete shape derivative of the same quantity under the vertex perturbation deltaP
void launch_constant_strain_load(const AsmArgs &a, const int32_t *elemNodes, const int32_t *dofForNode, const double *intGrad,
                                 const double *cstrain, const double *deltaP, double *out, hipStream_t s);
// deltaP != nullptr: strain(uNodes) + (delta strain)(uFixed)  (deltaAverageStrainField)
void launch_average_strain(const AsmArgs &a, const int32_t *elemNodes, const double *intGrad, const double *uNodes, double *out,
                           int wantStress, const double *uFixed, const double *deltaP, hipStream_t s, const double *addStrain = nullptr,
                           double *integral = nullptr);
void launch_apply_delta_K(const AsmArgs &a, const int32_t *elemNodes, const int32_t *dofForNode, const double *intGrad,
                          const double *uNodes, const double *deltaP, double *out, hipStream_t s);
// out[pair(ij<=kl)] += sum_e mutual energy (deltaP == nullptr) or its shape derivative; w: [flatLen][nNode][dim]
void launch_mutual_energies(const AsmArgs &a, const int32_t *elemNodes, const double *intGrad, const double *w, int64_t nNode,
                            const double *deltaP, double *out, hipStream_t s);
// strain / stress interpolant values per element: [nElem][1 (P1) | dim+1 (P2)][flatLen]
void launch_boundary_strain_field(const AsmArgs &a, const int32_t *elemNodes, const double *intGrad, int64_t nBE,
                                  const int32_t *bdryParent, const int32_t *bdryElemNodes, int npbe, const double *uNodes,
                                  int wantStress, double *out, hipStream_t s);
void launch_strain_field(const AsmArgs &a, const int32_t *elemNodes, const double *intGrad, const double *uNodes, int wantStress,
                         double *out, hipStream_t s);
// out[pair(ij<=kl)][nVert][dim] += d(mutual energy)/d(vertex position)
void launch_mutual_energy_differential(const AsmArgs &a, const int32_t *elemNodes, const double *intGrad, const double *w,
                                       int64_t nNode, int64_t nVert, double *out, hipStream_t s);
void launch_average_gradient(const AsmArgs &a, const int32_t *elemNodes, const double *intGrad, const double *uNodes, double *out,
                             hipStream_t s);

struct SpmvArgs {
    int dim;
    int64_t nChunk;
    const int32_t *chunkRow;
    const int32_t *rowPtr;
    const int32_t *colIdx;
    const double *vals;
    const float *vals32;                // non-null: the same tiled array rounded to FP32 (the linear level inside the multigrid preconditioner reads this
                                        // copy: the operator of a smoother tolerates it, and k_spmv is bound by exactly these bytes); products and sums stay FP64
    int chunkSlots;
    const uint8_t *fixedMask;           // per scalar row, may be null
    int xcd;                            // 1: XCD-contiguous chunk ranges (xcd_span)
    int pcgMode;                        // launch_spmv_nr with one vector: 0 = by ctl (none / Chronopoulos-Gear), 1 = classic PCG bookkeeping
    DetBuf det;                         // filled by the launcher (t_det)
};
// Matrix-free operator: y = K x without reading the assembled K. One lane per (element, local node i) pair:
// it evaluates the npe blocks K_e[i][j] in registers and applies them to the gathered x_j; pairs are grouped by
// row chunks (element-major inside a chunk) and reduced in LDS.
struct SpmvMfArgs {
    int dim, deg, npe, mat;
    int64_t nChunk;
    const int32_t *chunkRow;            // nChunk+1: first row of every chunk
    const int64_t *pairPtr;             // nChunk+1: first pair of every chunk
    const uint32_t *pairCode;           // e * npe + i
    const uint16_t *pairRow;            // row - chunkRow[chunk]
    const uint32_t *pairPos;            // inverse of pairCode: list position of pair e*npe+i (0xffffffff: not owned); null = element-major forces
    const int32_t *elemNodes;
    const int32_t *dofForNode;          // may be null
    const double *geo;
    int geoStride;
    const double *pairTable, *massTable;
    double pairConst[6];
    int maxRows;                        // rows per chunk (LDS accumulators)
    int xcd;                            // 1: XCD-contiguous chunk / element-group ranges (xcd_span)
    // cluster variant (k_mf_cluster); rowWrite restricts the writes of k_mf_rows to the interface rows
    int64_t clBlocks;
    int clMaxLocal, clBlockElems;
    const int32_t *clBlockPtr, *clEntryRow, *clEntryDest;
    const uint16_t *clLocalIdx;
    double *clIfaceBuf;
    const uint8_t *rowWrite;
    const int32_t *rowMap;              // k_mf_rows: chunk rows are indices into rowMap (null: identity)
    int64_t nElem;
    double *sig;                        // two-pass operator: nodal forces of every element, [nElem][npe][dim]
    const uint8_t *fixedMask;           // per scalar row, may be null
    const int32_t *clElemPtr;           // cluster variant: block b holds the elements [clElemPtr[b], clElemPtr[b+1]) of the operator's order (null: clBlockElems each)
    const int32_t *clElemPerm;          // cluster variant: original element of the operator's element e (null: identity); elemNodes is then in the new order
    int pcgMode;                        // launch_mf_*_nr with one vector: 0 = by ctl (none / Chronopoulos-Gear), 1 = classic PCG bookkeeping
    int clLaneStride;                   // lane t of a block takes element (t * stride) % blockElems (1: identity); coprime to blockElems
    const double *vertPos;              // cluster variant, constant material: corner positions [nVert][dim]; gradients recomputed (null: read the records)
    DetBuf det;                         // filled by the launcher (t_det); non-null = deterministic accumulation as well
    double shift[6];                    // launch_mf_cluster_constant_strain: the constant strain (flattened, TENSOR shear) added to grad u
};
// constantStrainLoad on the cluster operator's lists (LinearElasticity.hh:551-562): y = int (C : cstrain) grad phi_i = the operator's nodal forces
// for a field of constant strain `cstrain` -- the element routine with u = 0 and the strain added, summed through the same LDS accumulators
void launch_mf_cluster_constant_strain(const SpmvMfArgs &a, const double *cstrainFlat, double *y, hipStream_t s);
// out6 += sum over elements and edges of e e^T (xx, yy, zz, yz, xz, xy): the mesh's stretch for MFH_PRECOND_AUTO
void launch_edge_covariance(int64_t nElem, int dim, int npe, const int32_t *elemNodes, const double *vertPos, double *out6, hipStream_t s);
// neumannLoad on the device (LinearElasticity.hh:703-717): out[DoF(node)] += w[local node] * |b| * traction_b over the boundary elements (out is NOT zeroed here)
void launch_neumann_load(int64_t nBE, int npbe, int dim, const double *w6, const int32_t *bdryElemNodes, const int32_t *dofForNode, const double *bdryVol,
                         const double *traction, double *out, hipStream_t s);
void launch_spmv_mf(const SpmvMfArgs &a, const double *x, double *y, double *dotOut, double *scal, int it, const double *stopPtr,
                    bool pcg, hipStream_t s);
void launch_spmv_mf_cluster(const SpmvMfArgs &a, const double *x, double *y, double *dotOut, double *scal, int it, const double *stopPtr,
                            bool pcg, hipStream_t s);
void launch_spmv_mf2(const SpmvMfArgs &a, const double *x, double *y, double *dotOut, double *scal, int it, const double *stopPtr,
                     bool pcg, hipStream_t s);

// y = A x (optionally masked), optional dot accumulation: dotOut[0] += x_rows . y
void launch_spmv(const SpmvArgs &a, const double *x, double *y, double *dotOut, hipStream_t s);
// the same product from the upper-triangle storage (blocks (r, c >= r)): A^T x_r of every stored off-diagonal block added to y_c with global atomics
void launch_spmv_sym(const SpmvArgs &a, int64_t nRows, const double *x, double *y, double *dotOut, hipStream_t s);

void launch_untile_vals(int dim, int64_t nnzb, const double *tiled, double *aos, hipStream_t s);
void launch_extract_diag_inv(int dim, int64_t nRows, const int32_t *rowPtr, const int32_t *colIdx, const double *vals,
                             const uint8_t *fixedMask, int precondKind, double *dinv, hipStream_t s);
void launch_precond(int dim, int64_t nRows, const double *dinv, const double *r, double *z, hipStream_t s);

// PCG step kernels. scal: device array of per-iteration reductions, 4 doubles per iteration
// [r.z, p.Ap, r.r, unused], zero-filled once per solve. stopPtr[0] = rtol^2 * b.b: once
// scal[it].rr <= stop every kernel of the remaining iterations is a no-op.
void launch_pcg_init(int dim, int64_t nRows, const double *dinv, const double *b, double *x, double *r, double *z,
                     double *p, double *scal, hipStream_t s);
void launch_pcg_spmv(const SpmvArgs &a, const double *p, double *Ap, double *scal, int it, const double *stopPtr,
                     hipStream_t s);
void launch_pcg_update(int dim, int64_t nRows, const double *dinv, const double *Ap,
                       double *r, double *z, double *scal, int it, const double *stopPtr, hipStream_t s);
void launch_dev_update_xr(int64_t n, const double *num, const double *den, const double *p, const double *Ap, double *x, double *r, hipStream_t s);
void launch_dev_direction(int64_t n, const double *num, const double *den, const double *z, double *p, hipStream_t s);
void launch_dev_dots(int64_t n, const double *r, const double *z, double *out, hipStream_t s);
void launch_advance_base(double *stop, int n, hipStream_t s);   // stop[3] += n (iteration base of graph-captured PCG blocks)
void launch_pcg_direction(int64_t n, const double *z, double *p, double *x, const double *scal, int it, const double *stopPtr,
                          hipStream_t s);

struct TLArgs {
    int dim, nModes, nAgg;
    int64_t nDoF;
    const int32_t *aggOfDof;      // nDoF
    const double *relPos;         // nDoF x 3: (position - aggregate centroid) / H
    const uint8_t *fixedMask;     // per scalar variable, may be null
};
void launch_pcg_update_noz(int dim, int64_t nRows, const double *Ap, double *r, double *scal, int it,
                           const double *stopPtr, hipStream_t s);
void launch_tl_fill(const TLArgs &t, const int32_t *colorOfAgg, int color, int mode, double *v, hipStream_t s);
void launch_tl_restrict(const TLArgs &t, const int32_t *aggPtr, const int32_t *dofsByAgg, const double *w, double *rc, hipStream_t s);
void launch_tl_scatter(int nAgg, int nModes, int nColor, const int32_t *nbrOfColor, int color, int mode, const double *R, double *Ac,
                       hipStream_t s);
void launch_tl_rap(const TLArgs &t, int64_t nRows, const int32_t *rowPtr, const int32_t *colIdx, const double *vals, double *Ac,
                   hipStream_t s);
// (NR > 1: NR right-hand sides at once, vectors [aggregate][mode][NR])
void launch_st_spmv(int dim, int64_t nAgg, const int32_t *nbr, const double *A, const float *A32, const double *x, double *y, const double *scal, int it, const double *stop, hipStream_t s, int NR = 1);
void launch_st_dinv(int dim, int64_t nAgg, const double *A, double *Dinv, hipStream_t s);
void launch_st_cheb(int dim, int64_t nAgg, const double *Dinv, const double *rin, const double *t, double *rout, double *d, double *x, double a, double b,
                    bool first, bool assign, const double *scal, int it, const double *stop, hipStream_t s, int NR = 1);
void launch_st_rap(int dim, int64_t nParents, const int32_t *childPtr, const int32_t *childIdx, const int32_t *nbr, const double *A, const int32_t *parent,
                   const double *rel, const int32_t *coordC, double *Ac, const int *wrapNbC, hipStream_t s);
void launch_st_restrict(int dim, int64_t nParents, const int32_t *childPtr, const int32_t *childIdx, const double *rel, const double *r, const double *t, double *rc,
                        const double *scal, int it, const double *stop, hipStream_t s, int NR = 1);
void launch_st_prolong_add(int dim, int64_t nAgg, const int32_t *parent, const double *rel, const double *xc, double *x, double alpha, const double *scal, int it,
                           const double *stop, hipStream_t s, int NR = 1);
void launch_st_to_dense(int dim, int64_t nAgg, const int32_t *nbr, const double *A, double *Ad, hipStream_t s);
void launch_mg_zero(int64_t n, double *v, const double *scal, int it, const double *stop, hipStream_t s);
void launch_st_mirror_upper(double *stencil, const int32_t *nbr, int64_t nAgg, int dim, hipStream_t s);
void launch_tl_rap_agg(const TLArgs &t, const int32_t *aggPtr, const int32_t *dofsByAgg, const int32_t *binCoord, const int32_t *rowPtr,
                       const int32_t *colIdx, const double *vals, double *Ac, hipStream_t s, bool upperOnly = false, int64_t nOwnedRows = 0,
                       double *stencil = nullptr, int *farCount = nullptr, const int *wrapNb = nullptr);
bool dense_spd_inverse_device(double *A, double *X, double *Ainv, double *Dinv, int64_t mp, int *notSpdDev, hipStream_t s);
void launch_tl_gemv(int64_t m, int64_t ld, const double *A, const double *x, double *y, hipStream_t s);
void launch_tl_prep(int64_t m, int64_t mp, const double *Ac, const uint8_t *dead, double maxd, double *Ap, hipStream_t s);
void launch_tl_apply(const TLArgs &t, const double *dinv, const double *r, const double *yc, double *z, double *scal, int it,
                     const double *stopPtr, hipStream_t s);

void launch_axpby(int64_t n, double a, const double *x, double b, double *y, hipStream_t s); // y = a x + b y
void launch_mask(int64_t n, const uint8_t *mask, double *v, hipStream_t s);                   // v[mask]=0
void launch_scatter_values(int64_t n, const int64_t *idx, const double *val, double *v, int64_t bound, hipStream_t s);
void launch_dot(int64_t n, const double *a, const double *b, double *out, hipStream_t s);    // *out += a.b


// ---- batched / distributed Chronopoulos-Gear PCG (mfh_solver.cpp): NR interleaved vectors, entry ((row NR + k) dim + c)
bool op_batch_supported(int dim, int nr);
void launch_mf_cluster_nr(const SpmvMfArgs &a, int NR, const double *x, double *y, double *dotOut, double *scal, int it, const double *ctl,
                          const int32_t *blockList, int64_t nList, hipStream_t s);
void launch_mf_rows_nr(const SpmvMfArgs &a, int NR, const double *x, double *y, double *dotOut, double *scal, int it, const double *ctl, hipStream_t s);
void launch_spmv_nr(const SpmvArgs &a, int NR, const double *x, double *y, double *dotOut, double *scal, int it, const double *ctl,
                    const int32_t *chunkList, int64_t nList, hipStream_t s);
void launch_flag_halo_blocks(int64_t nBlocks, const int32_t *blockPtr, const int32_t *entryDest, uint8_t *flag, hipStream_t s);
void launch_flag_halo_chunks(int64_t nChunk, const int32_t *chunkRow, const int32_t *rowPtr, const int32_t *colIdx, int64_t nOwnedCols, uint8_t *flag,
                             hipStream_t s);
void launch_cg_update(int dim, int64_t nRows, int NR, const double *dinv, double *u, const double *w, double *p, double *sv, double *x, double *r,
                      double *scal, int it, const double *ctl, bool skipU, hipStream_t s);
void launch_cg_init(int dim, int64_t nRows, int NR, const double *dinv, const double *r, double *u, double *scal, bool skipU, hipStream_t s);
void launch_tl_restrict_nr(const TLArgs &t, int NR, const int32_t *aggPtr, const int32_t *dofsByAgg, const double *w, double *rc, hipStream_t s);
void launch_tl_gemv_nr(int64_t m, int64_t ld, int NR, const double *A, const double *x, double *y, hipStream_t s);
void launch_tl_apply_nr(const TLArgs &t, int NR, const double *dinv, const double *r, const double *yc, double *z, double *scal, int it,
                        const double *ctl, hipStream_t s);
extern int g_vecGridCap;
// p-multigrid (mfh_multigrid.cpp); scal / it / stop: the gate of the PCG iteration (null: none) -- of ONE classic loop by default; inside a
// GateScope(NR, stride) of NR loops advancing in lockstep (loop k: history scal + k stride, control block stop + 4 k), closed when all are.
// NR > 1 on the linear level and below: NR interleaved right-hand sides (entry ((row NR + k) dim + c); coarse vectors [coarse index][NR]).
extern thread_local int t_gateNr;
extern thread_local int64_t t_gateStride;
struct GateScope {
    int saved;
    int64_t savedStride;
    explicit GateScope(int nr, int64_t stride = 0) : saved(t_gateNr), savedStride(t_gateStride) { t_gateNr = nr; t_gateStride = stride; }
    ~GateScope() { t_gateNr = saved; t_gateStride = savedStride; }
};
void launch_mg_cheb(int dim, int64_t nRows, const double *dinv, const double *rin, const double *t, double *rout, double *d, double *x,
                    double a, double b, bool first, bool assign, const double *scal, int it, const double *stop, hipStream_t s, int NR = 1);
// NR > 1: fine vectors separate (fineStride doubles apart), coarse result interleaved
void launch_mg_restrict(int dim, int64_t nCoarse, const int32_t *fineOf, const int32_t *resPtr, const int32_t *resIdx, const double *r, const double *t,
                        const uint8_t *coarseMask, double *rc, const double *scal, int it, const double *stop, hipStream_t s, int NR = 1, int64_t fineStride = 0);
// ldc: doubles between coarse rows (0 = dim; NR dim with xc pointing at vector k of NR interleaved coarse vectors)
void launch_mg_prolong_add(int dim, int64_t nFine, const int32_t *parA, const int32_t *parB, const double *xc, const uint8_t *fineMask, double *x,
                           const double *scal, int it, const double *stop, hipStream_t s, int ldc = 0);
// all NR right-hand sides of the batched V-cycle in one launch: xc interleaved, x separate (vector k at x + k vecStride), gated per loop
void launch_mg_prolong_add_nr(int dim, int NR, int64_t nFine, const int32_t *parA, const int32_t *parB, const double *xc, const uint8_t *fineMask, double *x,
                              int64_t vecStride, const double *scal, int64_t scalStride, int it, const double *stop, hipStream_t s);
void launch_mg_tl_prolong_add(const TLArgs &t, const double *yc, double *x, double alpha, const double *scal, int it, const double *stop, hipStream_t s, int NR = 1);
void launch_fill_hash(int64_t n, double *v, hipStream_t s);
void launch_to_f32(int64_t n, const double *src, float *dst, hipStream_t s);
void launch_take_columns_i32(int64_t n, int W, int w, const int32_t *src, int32_t *dst, hipStream_t s);
void launch_mg_diff(int64_t n, const double *a, const double *b, double *out, const double *scal, int it, const double *stop, hipStream_t s);
void launch_mg_rz(int64_t n, const double *r, double *z, const uint8_t *mask, double *scalOut, int it, const double *scal, const double *stop, hipStream_t s);
void launch_mg_cheb_rz(int dim, int64_t nRows, const double *dinv, const float *dinv32, const double *rin, const double *t, double *x, double b, const uint8_t *mask,
                       double *scalOut, int it, const double *scal, const double *stop, hipStream_t s);
void launch_pcg_update_presmooth(int dim, int64_t nRows, const double *dinv, const float *dinv32, const double *Ap, double *r, double *z, double zs, double *scal, int it,
                                 const double *stopPtr, hipStream_t s);
void launch_add_scalar(double *p, double v, hipStream_t s);   // *p += v
void launch_pack_rows(int64_t n, int W, const int32_t *idx, const double *src, double *dst, hipStream_t s);
void launch_unpack_add_rows(int64_t n, int W, const int32_t *idx, const double *src, double *dst, hipStream_t s);   // dst[idx[j]][:] += src[j][:]
void launch_pack_rows_f32(int64_t n, int W, const int32_t *idx, const float *src, float *dst, hipStream_t s);
void launch_remap_i32(int64_t n, const int32_t *map, int32_t *v, hipStream_t s);                                  // v[k] = map[v[k]] where v[k] >= 0
void launch_interleave(int64_t nRows, int NR, int dim, const double *src, double *dst, bool toInterleaved, int64_t sepStride, hipStream_t s);
void launch_norms_nr(int64_t nRows, int NR, int dim, const double *v, double *out, hipStream_t s);
void launch_mask_nr(int64_t nRows, int NR, int dim, const uint8_t *mask, double *v, hipStream_t s);
void launch_scatter_values_nr(int64_t n, int NR, int dim, const int64_t *idx, const double *val, double *v, int64_t rowBound, hipStream_t s);

}} // namespace mfh::k
