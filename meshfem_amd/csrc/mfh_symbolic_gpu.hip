// Symbolic phase on the device: the block-CSR pattern of K and the element-major gather lists,
// produced by two radix sorts over the nElem*npe^2 (row, col, code) contributions instead of the
// threaded per-row sorts of mfh_symbolic.cpp (identical output, bit for bit; ~10x faster at 5 M P2
// tets and independent of the host's core count). This is the once-per-mesh part of the reference's
// TripletMatrix::sumRepeated (SparseMatrices.hh:280-374): sort by (col,row), merge duplicates.
// rocPRIM (header-only) provides the device-wide radix sort / scan primitives.
#include "mfh_internal.hh"
#include <rocprim/rocprim.hpp>

namespace mfh {

namespace {

// Sort keys of the contributions: (row, col, ij) = ((row 2^cb + col) << 7) | ij with cb = bits of the column count and ij = i npe + j
// the position inside the element matrix; the value carried along is the ELEMENT. The sort looks at the bits above ij only, so equal
// (row, col) keep their element order; element and ij together are the gather code. Nothing here multiplies the element count by
// npe^2: the 2^32 / 100 = 42.9 M ceiling on quadratic tets per device is gone (what remains: 2^32 contributions after the
// owned-row / upper-triangle filter, i.e. 78 M quadratic tets with upper-triangle storage, and rb + cb <= 57 bits of row and column).
constexpr int SYM_IJ_BITS = 7;
__device__ __host__ inline uint64_t sym_key(uint64_t row, uint64_t col, unsigned ij, unsigned cb) { return (((row << cb) | col) << SYM_IJ_BITS) | ij; }
__device__ inline int64_t sym_row(uint64_t key, unsigned cb) { return (int64_t)(key >> (cb + SYM_IJ_BITS)); }
__device__ inline int64_t sym_col(uint64_t key, unsigned cb) { return (int64_t)((key >> SYM_IJ_BITS) & ((1ull << cb) - 1)); }

// The same keys, only the valid ones (owned row; with upper-only storage: col >= row), in code order: pass 1 counts them per element,
// an exclusive scan gives every element its offset, pass 2 writes. The sort then runs over nC instead of nElem npe^2 entries
// (55 % of them with the upper triangle of a quadratic mesh).
__global__ void __launch_bounds__(256) k_sym_count(int64_t nElem, int npe, const int32_t *__restrict__ elemNodes, const int32_t *__restrict__ dofForNode,
                                                   int64_t nRows, int upperOnly, uint32_t *__restrict__ cnt) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nElem; e += (int64_t)gridDim.x * 256) {
        int64_t dof[10];
        for (int i = 0; i < npe; ++i) { const int32_t n = elemNodes[e * npe + i]; dof[i] = dofForNode ? dofForNode[n] : n; }
        uint32_t c = 0;
        for (int i = 0; i < npe; ++i) {
            if (dof[i] >= nRows) continue;
            if (!upperOnly) { c += (uint32_t)npe; continue; }
            for (int j = 0; j < npe; ++j) c += dof[j] >= dof[i];
        }
        cnt[e] = c;
    }
}
__global__ void __launch_bounds__(256) k_sym_gen_compact(int64_t nElem, int npe, const int32_t *__restrict__ elemNodes, const int32_t *__restrict__ dofForNode,
                                                         int64_t nRows, int upperOnly, const uint32_t *__restrict__ off, uint64_t *__restrict__ key,
                                                         uint32_t *__restrict__ val, unsigned cb) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nElem; e += (int64_t)gridDim.x * 256) {
        int64_t dof[10];
        for (int i = 0; i < npe; ++i) { const int32_t n = elemNodes[e * npe + i]; dof[i] = dofForNode ? dofForNode[n] : n; }
        int64_t w = off[e];
        for (int i = 0; i < npe; ++i) {
            if (dof[i] >= nRows) continue;
            for (int j = 0; j < npe; ++j) {
                if (upperOnly && dof[j] < dof[i]) continue;
                key[w] = sym_key((uint64_t)dof[i], (uint64_t)dof[j], (unsigned)(i * npe + j), cb);
                val[w] = (uint32_t)e;
                ++w;
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_sym_heads(int64_t n, const uint64_t *__restrict__ key, uint32_t *__restrict__ head, unsigned shift = 0) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        head[k] = (k == 0 || (key[k] >> shift) != (key[k - 1] >> shift)) ? 1u : 0u;
}

// at slot heads: column index and row length; at row heads: first contribution of the row
__global__ void __launch_bounds__(256) k_sym_pattern(int64_t n, const uint64_t *__restrict__ key, const uint32_t *__restrict__ slotP1,
                                                     int32_t *__restrict__ colIdx, int32_t *__restrict__ rowLen,
                                                     int64_t *__restrict__ rowCStart, int64_t nRows, unsigned long long *__restrict__ nMirror, unsigned cb) {
    unsigned long long mirror = 0;      // summed over the wave before it reaches the one global counter (59-98 ms -> a few ms at config 3:
                                        // 50 M single-address atomics were the whole kernel)
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const uint64_t kk = key[k] >> SYM_IJ_BITS;                    // (row, col) without the position inside the element matrix
        const uint64_t prev = k == 0 ? ~kk : key[k - 1] >> SYM_IJ_BITS;
        if (kk == prev) continue;                                     // not a slot head
        const int64_t row = sym_row(key[k], cb);
        const int64_t col = sym_col(key[k], cb);
        colIdx[slotP1[k] - 1] = (int32_t)col;
        atomicAdd(&rowLen[row], 1);
        if (col > row && col < nRows) ++mirror;                       // blocks whose transpose is a block of K too (Symbolic::nMirror)
        if (k == 0 || sym_row(key[k - 1], cb) != row) rowCStart[row] = k;
    }
    for (int off = 32; off > 0; off >>= 1) mirror += __shfl_down(mirror, off, 64);
    if ((threadIdx.x & 63) == 0 && mirror) atomicAdd(nMirror, mirror);
}

// first contribution of every chunk = that of its first row (an empty row takes the next row's: none in a valid mesh)
__global__ void __launch_bounds__(256) k_sym_chunk_starts(int64_t nChunk, const int32_t *__restrict__ chunkRow, const int64_t *__restrict__ rowCStart,
                                                          int64_t nRows, int64_t nC, int64_t *__restrict__ contribPtr) {
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c <= nChunk; c += (int64_t)gridDim.x * 256) {
        int64_t v = nC;
        if (c < nChunk)
            for (int64_t r = chunkRow[c]; r < nRows; ++r)
                if (rowCStart[r] >= 0) { v = rowCStart[r]; break; }
        contribPtr[c] = v;
    }
}

// chunk of every row and first slot of every chunk, from the chunks' first rows (filled on the device: the per-row table never exists on the host)
__global__ void __launch_bounds__(256) k_sym_chunk_tables(int64_t nChunk, const int32_t *__restrict__ chunkRow, const int32_t *__restrict__ rowPtr,
                                                          int32_t *__restrict__ chunkOfRow, int32_t *__restrict__ chunkBase) {
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nChunk; c += (int64_t)gridDim.x * 256) {
        const int32_t r0 = chunkRow[c], r1 = chunkRow[c + 1];
        chunkBase[c] = rowPtr[r0];
        for (int32_t r = r0; r < r1; ++r) chunkOfRow[r] = (int32_t)c;
    }
}

// smallest element a chunk gathers from (one wave per chunk); flag[0] is raised when a chunk's elements span 2^25 or more: the
// chunk-relative packed code (see k_assemble_gather) has 25 bits for the element
__global__ void __launch_bounds__(64) k_sym_chunk_elem_base(int64_t nChunk, const int64_t *__restrict__ contribPtr, const uint32_t *__restrict__ elem,
                                                            int32_t *__restrict__ chunkElemBase, int *flag, int32_t *__restrict__ offenders, int maxOffenders) {
    const int64_t b = blockIdx.x;
    const int64_t kb = contribPtr[b], ke = contribPtr[b + 1];
    uint32_t lo = 0xffffffffu, hi = 0;
    for (int64_t k = kb + threadIdx.x; k < ke; k += 64) {
        const uint32_t e = elem[k];
        lo = e < lo ? e : lo;
        hi = e > hi ? e : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t l2 = __shfl_xor(lo, off, 64), h2 = __shfl_xor(hi, off, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (ke <= kb) lo = 0;
    if (threadIdx.x == 0) {
        chunkElemBase[b] = (int32_t)lo;
        if (ke > kb && hi - lo >= (1u << (32 - SYM_IJ_BITS))) {
            const int q = atomicAdd(flag, 1);                        // flag[0] = number of offending chunks
            if (q < maxOffenders) offenders[q] = (int32_t)b;
        }
    }
}
// element of the first contribution of every row of the offending chunks (a representative of where the row gathers from)
__global__ void __launch_bounds__(256) k_sym_row_rep(int nOff, const int32_t *__restrict__ offenders, const int32_t *__restrict__ chunkRow,
                                                     const int64_t *__restrict__ rowCStart, const uint32_t *__restrict__ elem, int stride,
                                                     uint32_t *__restrict__ out) {
    const int o = blockIdx.x;
    if (o >= nOff) return;
    const int32_t c = offenders[o];
    const int32_t r0 = chunkRow[c], r1 = chunkRow[c + 1];
    for (int r = r0 + (int)threadIdx.x; r < r1 && r - r0 < stride; r += 256) {
        const int64_t k = rowCStart[r];
        out[(int64_t)o * stride + (r - r0)] = k >= 0 ? elem[k] : 0xffffffffu;
    }
}

// per contribution: its slot inside the chunk and its gather code -- chunk-relative and packed, ((element - chunk's first element) << 7) | ij
// (chunkElemBase != null), or absolute, element npe^2 + ij. Either orders a chunk's contributions element-major.
__global__ void __launch_bounds__(256) k_sym_key2(int64_t n, const uint64_t *__restrict__ key, const uint32_t *__restrict__ elem,
                                                  const uint32_t *__restrict__ slotP1, const int32_t *__restrict__ chunkOfRow,
                                                  const int32_t *__restrict__ chunkBase, const int32_t *__restrict__ chunkElemBase, unsigned cb, unsigned npe2,
                                                  uint32_t *__restrict__ code, uint16_t *__restrict__ lslot, int32_t *__restrict__ scatterSlot) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const int64_t row = sym_row(key[k], cb);
        const int32_t ch = chunkOfRow[row];
        const int32_t slot = (int32_t)(slotP1[k] - 1);
        const uint32_t ij = (uint32_t)(key[k] & ((1u << SYM_IJ_BITS) - 1)), e = elem[k];
        code[k] = chunkElemBase ? (((e - (uint32_t)chunkElemBase[ch]) << SYM_IJ_BITS) | ij) : e * npe2 + ij;
        lslot[k] = (uint16_t)(slot - chunkBase[ch]);
        if (scatterSlot) scatterSlot[(int64_t)e * npe2 + ij] = slot;
    }
}

__global__ void __launch_bounds__(256) k_sym_codes(int64_t n, const uint64_t *__restrict__ key2, uint32_t *__restrict__ code) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) code[k] = (uint32_t)(key2[k] & 0xffffffffu);
}

inline int grid_of(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 65536)); }
inline unsigned bits_for(uint64_t v) { unsigned b = 1; while ((v >> b) != 0 && b < 32) ++b; return b; }

#define RP(expr)                                                                                       \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) throw mfh::Error(MFH_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Greedy row chunks: every chunk takes as many whole rows as fit into chunkSlots slots. breaks: rows (ascending) at which a chunk must end.
// The scan is sequential by nature (a chunk starts where the previous one ended), but chains started at different rows fall into step as
// soon as they share one boundary: the host threads each run the scan from the start of their own range of rows, and a stitching pass follows
// the true chain into every range only until it meets a boundary that range's own scan produced (a few chunks), then adopts the rest.
// The result is the sequential one (57.6 M rows at 119^3: 2 x 55 ms of the first assembly before).
namespace {
inline int64_t greedy_step(const int32_t *rowPtr, int64_t nRows, int chunkSlots, const std::vector<int64_t> &breaks, int64_t r) {
    const int32_t s0 = rowPtr[r];
    const auto it = std::upper_bound(breaks.begin(), breaks.end(), r);
    const int64_t stop = it != breaks.end() ? *it : -1;
    int64_t r2 = r + 1;
    while (r2 < nRows && r2 != stop && rowPtr[r2 + 1] - s0 <= chunkSlots) ++r2;
    return r2;
}
}   // namespace

} // namespace

std::vector<int32_t> make_chunks(const int32_t *rowPtr, int64_t nRows, int chunkSlots, const std::vector<int64_t> &breaks, int64_t grain,
                                 int maxThreads) {
    std::vector<int32_t> chunkRow{0};
    const int nt = (int)std::min<int64_t>(maxThreads > 0 ? maxThreads : host_threads(), nRows / std::max<int64_t>(1, grain));
    if (nt <= 1) {
        for (int64_t r = 0; r < nRows;) { r = greedy_step(rowPtr, nRows, chunkSlots, breaks, r); chunkRow.push_back((int32_t)r); }
        return chunkRow;
    }
    std::vector<std::vector<int32_t>> part((size_t)nt);          // boundaries of the scan started at the range's first row (the first row itself excluded)
    auto rangeStart = [&](int t) { return nRows * t / nt; };
    parallel_ranges(nt, [&](int64_t tb, int64_t te, int) {
        for (int64_t t = tb; t < te; ++t) {
            const int64_t end = rangeStart((int)t + 1);
            std::vector<int32_t> &L = part[(size_t)t];
            L.reserve((size_t)((end - rangeStart((int)t)) / 8 + 16));
            for (int64_t r = rangeStart((int)t); r < end;) { r = greedy_step(rowPtr, nRows, chunkSlots, breaks, r); L.push_back((int32_t)r); }
        }
    }, 1);
    size_t total = 1;
    for (auto &L : part) total += L.size();
    chunkRow.reserve(total);
    int64_t cur = 0;                                            // end of the last chunk of the true chain
    for (int t = 0; t < nt; ++t) {
        const int64_t start = rangeStart(t), end = rangeStart(t + 1);
        const std::vector<int32_t> &L = part[(size_t)t];
        if (cur >= end) continue;                               // (a chunk that spans a whole range: not with ranges of 2^18 rows, but harmless)
        if (cur == start) { chunkRow.insert(chunkRow.end(), L.begin(), L.end()); cur = L.back(); continue; }
        if (L.empty()) continue;
        while (cur < end) {
            const auto it = std::lower_bound(L.begin(), L.end(), (int32_t)cur);
            if (it != L.end() && *it == (int32_t)cur) { chunkRow.insert(chunkRow.end(), it + 1, L.end()); cur = L.back(); break; }
            cur = greedy_step(rowPtr, nRows, chunkSlots, breaks, cur);
            chunkRow.push_back((int32_t)cur);
        }
    }
    return chunkRow;
}

namespace {

// Morton key of an element's centroid inside the bounding box (21 bits per axis in 3D, 31 in 2D)
__device__ inline uint64_t spread3(uint64_t v) {   // 21 bits -> every third bit
    v &= 0x1fffff;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__device__ inline uint64_t spread2(uint64_t v) {   // 31 bits -> every second bit
    v &= 0x7fffffff;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
__global__ void __launch_bounds__(256) k_elem_morton(int64_t nElem, int npe, int dim, const int32_t *__restrict__ elemNodes, const double *__restrict__ pos,
                                                    double lx, double ly, double lz, double sx, double sy, double sz, uint64_t *__restrict__ key,
                                                    uint32_t *__restrict__ val) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nElem; e += (int64_t)gridDim.x * 256) {
        double c[3] = {0, 0, 0};
        for (int k = 0; k <= dim; ++k) {
            const int64_t v = elemNodes[e * npe + k];
            for (int a = 0; a < dim; ++a) c[a] += pos[v * dim + a];
        }
        const double inv = 1.0 / (dim + 1);
        const double qx = (c[0] * inv - lx) * sx, qy = (c[1] * inv - ly) * sy, qz = dim == 3 ? (c[2] * inv - lz) * sz : 0.0;
        const uint64_t top = dim == 3 ? 0x1fffffull : 0x7fffffffull;
        auto q = [&](double t) { return (uint64_t)fmin(fmax(t, 0.0), (double)top); };
        key[e] = dim == 3 ? (spread3(q(qx)) | (spread3(q(qy)) << 1) | (spread3(q(qz)) << 2)) : (spread2(q(qx)) | (spread2(q(qy)) << 1));
        val[e] = (uint32_t)e;
    }
}
__global__ void __launch_bounds__(256) k_cell_starts(int64_t n, const uint64_t *__restrict__ key, const uint32_t *__restrict__ cellP1,
                                                     int32_t *__restrict__ cellStart) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        if (k == 0 || key[k] != key[k - 1]) cellStart[cellP1[k] - 1] = (int32_t)k;
}
__global__ void __launch_bounds__(256) k_permute_rows_i32(int64_t n, int W, const uint32_t *__restrict__ perm, const int32_t *__restrict__ src,
                                                         int32_t *__restrict__ dst, int32_t *__restrict__ permOut) {
    const int64_t total = n * W;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t e = k / W;
        const int c = (int)(k - e * W);
        dst[k] = src[(int64_t)perm[e] * W + c];
        if (c == 0) permOut[e] = (int32_t)perm[e];
    }
}
} // namespace

// Device symbolic phase (element-major gather lists). Host-side S receives rowPtr, chunk tables and
// contribPtr; colIdx and the gather lists stay on the device (downloaded on demand by the API layer).
void build_symbolic_device(const HostMesh &m, const int32_t *dElemNodes, const int32_t *dDofForNode, int64_t nDoF, int64_t nOwnedDoF,
                           int chunkSlots, bool wantScatter, hipStream_t s, Symbolic &S, DBuf<int32_t> &dRowPtr, DBuf<int32_t> &dColIdx,
                           DBuf<uint32_t> &dContribCode, DBuf<uint16_t> &dContribSlot, DBuf<int32_t> &dScatter, bool upperOnly,
                           DBuf<int32_t> *dChunkElemBase, bool *codesPacked) {
    // dChunkElemBase != null: the gather codes come out chunk-relative and packed (what k_assemble_gather reads) whenever every chunk's
    // elements span less than 2^25; *codesPacked says which. Absolute codes (element npe^2 + ij, the format of the host lists) and the
    // element -> slot scatter map of the atomic variant need nElem npe^2 < 2^32.
    const int npe = m.npe;
    const int64_t N = m.nElem * npe * npe;
    const bool fitsAbsolute = (double)N < 4294967295.0;
    if (codesPacked) *codesPacked = false;
    if (!fitsAbsolute && (wantScatter || !dChunkElemBase))
        throw Error(MFH_ERR_UNSUPPORTED, "mesh too large for 32-bit absolute contribution codes: needs the packed gather codes (option asm_packed_codes 1, no "
                                         "host copy of the lists, gather assembly)");
    if ((double)m.nElem >= 4294967295.0) throw Error(MFH_ERR_UNSUPPORTED, "more than 2^32 elements on one device");
    const unsigned cb = bits_for((uint64_t)std::max<int64_t>(nDoF, 1));
    if (bits_for((uint64_t)std::max<int64_t>(nOwnedDoF, 1)) + cb + SYM_IJ_BITS > 64 || npe * npe > (1 << SYM_IJ_BITS))
        throw Error(MFH_ERR_UNSUPPORTED, "row and column counts do not fit the 64-bit sort keys of the symbolic phase");
    S = Symbolic();
    S.nRows = nOwnedDoF;
    S.nCols = nDoF;
    const int64_t nRows = S.nRows;
    const bool timing = getenv("MFH_SYM_TIMING") != nullptr;
    double tp = now_ms();
    // every phase ends with a stream synchronisation: the buffers released and allocated in between then never meet kernels in
    // flight (hipFree / hipMalloc behind a busy stream stalled for 0.2 - 0.8 s, erratically, at 5 M elements)
    auto lap = [&](const char *what) {
        (void)hipStreamSynchronize(s);
        if (!timing) return;
        const double t = now_ms();
        fprintf(stderr, "[symbolic] %-36s %8.2f ms\n", what, t - tp);
        tp = t;
    };

    DBuf<uint64_t> keyA, keyB;
    DBuf<uint32_t> valA, valB, slotP1;
    // ---- the valid contributions only, in code order (count per element, scan, write)
    DBuf<uint32_t> cnt;
    cnt.alloc((size_t)m.nElem + 1);
    MFH_HIP(hipMemsetAsync(cnt.p + m.nElem, 0, sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_sym_count, dim3(grid_of(m.nElem)), dim3(256), 0, s, m.nElem, npe, dElemNodes, dDofForNode, nRows, upperOnly ? 1 : 0, cnt.p);
    RP(hipGetLastError());
    DBuf<char> tmp;
    {
        size_t scanBytes0 = 0;
        RP(rocprim::exclusive_scan(nullptr, scanBytes0, cnt.p, cnt.p, (uint32_t)0, (size_t)m.nElem + 1, rocprim::plus<uint32_t>(), s));
        tmp.alloc(scanBytes0 + 16);
        RP(rocprim::exclusive_scan(tmp.p, scanBytes0, cnt.p, cnt.p, (uint32_t)0, (size_t)m.nElem + 1, rocprim::plus<uint32_t>(), s));
    }
    uint32_t nValidU = 0;
    MFH_HIP(hipMemcpyAsync(&nValidU, cnt.p + m.nElem, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    const int64_t nC = (int64_t)nValidU;                 // contributions whose row is owned (and, with upper-only storage, col >= row)
    if (nC == 0) throw Error(MFH_ERR_INVALID, "no element touches an owned row");
    keyA.alloc(nC); keyB.alloc(nC); valA.alloc(nC); valB.alloc(nC);
    hipLaunchKernelGGL(k_sym_gen_compact, dim3(grid_of(m.nElem)), dim3(256), 0, s, m.nElem, npe, dElemNodes, dDofForNode, nRows, upperOnly ? 1 : 0, cnt.p,
                       keyA.p, valA.p, cb);
    RP(hipGetLastError());
    cnt.release();

    lap("allocate + generate keys");
    // ---- sort by (row, col); stable, so equal keys stay in code order
    const unsigned beginBit1 = SYM_IJ_BITS, endBit1 = SYM_IJ_BITS + cb + bits_for((uint64_t)nRows);   // (row, col) only: ij rides along
    size_t tmpBytes = 0;
    RP(rocprim::radix_sort_pairs(nullptr, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)nC, beginBit1, endBit1, s));
    {
        const double tA = now_ms();
        if (tmpBytes + 16 > tmp.n) tmp.alloc(tmpBytes + 16);
        const double tB = now_ms();
        RP(rocprim::radix_sort_pairs(tmp.p, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)nC, beginBit1, endBit1, s));
        if (timing) {
            const double tC = now_ms();
            (void)hipStreamSynchronize(s);
            fprintf(stderr, "[symbolic] first sort: %.1f MB temporary, allocation %.2f ms, enqueue %.2f ms, run %.2f ms\n", tmpBytes / 1e6, tB - tA, tC - tB, now_ms() - tC);
        }
    }

    lap("sort by (row, col)");
    // ---- slots = distinct (row, col) pairs
    slotP1.alloc(nC);
    valA.release();
    hipLaunchKernelGGL(k_sym_heads, dim3(grid_of(nC)), dim3(256), 0, s, nC, keyB.p, slotP1.p, (unsigned)SYM_IJ_BITS);
    size_t scanBytes = 0;
    RP(rocprim::inclusive_scan(nullptr, scanBytes, slotP1.p, slotP1.p, (size_t)nC, rocprim::plus<uint32_t>(), s));
    if (scanBytes + 16 > tmp.n) tmp.alloc(scanBytes + 16);
    RP(rocprim::inclusive_scan(tmp.p, scanBytes, slotP1.p, slotP1.p, (size_t)nC, rocprim::plus<uint32_t>(), s));
    uint32_t nnzbU = 0;
    MFH_HIP(hipMemcpyAsync(&nnzbU, slotP1.p + (nC - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    if (nnzbU > 2147483647u) throw Error(MFH_ERR_UNSUPPORTED, "more than 2^31 blocks on one device");
    S.nnzb = nnzbU;

    lap("slot heads + scan");
    dColIdx.alloc((size_t)S.nnzb);
    DBuf<int32_t> rowLen;
    DBuf<int64_t> rowCStart;
    rowLen.alloc((size_t)nRows + 1);
    rowLen.zero(s);
    rowCStart.alloc((size_t)nRows + 1);
    MFH_HIP(hipMemsetAsync(rowCStart.p, 0xff, (size_t)(nRows + 1) * sizeof(int64_t), s));   // -1 = empty row
    DBuf<unsigned long long> dCount;
    dCount.alloc(1);
    dCount.zero(s);
    hipLaunchKernelGGL(k_sym_pattern, dim3(grid_of(nC)), dim3(256), 0, s, nC, keyB.p, slotP1.p, dColIdx.p, rowLen.p, rowCStart.p, nRows, dCount.p, cb);
    {
        unsigned long long nm = 0;
        dCount.download(&nm, 1, s);
        S.nMirror = (int64_t)nm;
    }
    RP(hipGetLastError());
    dRowPtr.alloc((size_t)nRows + 1);
    size_t exBytes = 0;
    RP(rocprim::exclusive_scan(nullptr, exBytes, rowLen.p, dRowPtr.p, (int32_t)0, (size_t)nRows + 1, rocprim::plus<int32_t>(), s));
    if (exBytes + 16 > tmp.n) tmp.alloc(exBytes + 16);
    RP(rocprim::exclusive_scan(tmp.p, exBytes, rowLen.p, dRowPtr.p, (int32_t)0, (size_t)nRows + 1, rocprim::plus<int32_t>(), s));
    resize_prefaulted(S.rowPtr, (size_t)nRows + 1);      // (page faults on all host threads while the scan runs on the device)
    dRowPtr.download(S.rowPtr.data(), S.rowPtr.size(), s);
    {
        std::vector<int32_t> mx((size_t)host_threads() + 1, 0);
        parallel_ranges(nRows, [&](int64_t b, int64_t e, int tid) {
            int32_t v = 0;
            for (int64_t r = b; r < e; ++r) v = std::max(v, S.rowPtr[r + 1] - S.rowPtr[r]);
            mx[(size_t)tid] = std::max(mx[(size_t)tid], v);
        });
        for (int32_t v : mx) S.maxRowLen = std::max(S.maxRowLen, v);
    }

    lap("pattern + row pointers + downloads");
    // ---- chunks (host: a scan over the row pointers)
    if (chunkSlots < 64) chunkSlots = 64;
    if (S.maxRowLen > chunkSlots) chunkSlots = ((S.maxRowLen + 63) / 64) * 64;
    if (chunkSlots > 2048)
        throw Error(MFH_ERR_UNSUPPORTED, "a block row has more than 2048 blocks (vertex valence too high for LDS accumulation)");
    S.chunkSlots = chunkSlots;
    S.spmvChunkSlots = std::max(512, chunkSlots);
    std::thread spmvChunks([&]() { S.spmvChunkRow = make_chunks(S.rowPtr.data(), nRows, S.spmvChunkSlots); });   // (the greedy scans are sequential by nature)
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{spmvChunks};          // also when a step below throws
    // The gather codes are packed relative to the smallest element of their chunk (25 bits). On meshes of 2^25 elements and more a chunk
    // that holds rows from two ends of the node numbering -- the last corner vertices and the first cell centres of the generator's
    // meshes, the last vertex rows and the first edge-node rows -- gathers from elements further apart than that: such chunks are found
    // (k_sym_chunk_elem_base), cut where the rows' first elements jump, and the chunk tables rebuilt (a handful of chunks, two rounds).
    std::vector<int64_t> breaks;
    int64_t nChunk = 0;
    DBuf<int64_t> dSeg;                            // the chunks' first contributions: gathered on the device (the per-row table stays there)
    DBuf<int32_t> dCR;
    bool packed = false;
    for (int round = 0; round < 4; ++round) {
        S.chunkRow = make_chunks(S.rowPtr.data(), nRows, chunkSlots, breaks);
        lap("  chunks: greedy scan");
        nChunk = S.nChunk();
        S.contribPtr.resize((size_t)nChunk + 1);
        dCR.upload(S.chunkRow, s);
        dSeg.alloc((size_t)nChunk + 1);
        hipLaunchKernelGGL(k_sym_chunk_starts, dim3(grid_of(nChunk + 1)), dim3(256), 0, s, nChunk, dCR.p, rowCStart.p, nRows, nC, dSeg.p);
        RP(hipGetLastError());
        MFH_HIP(hipMemcpyAsync(S.contribPtr.data(), dSeg.p, (size_t)(nChunk + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        MFH_HIP(hipStreamSynchronize(s));
        lap("  chunks: upload + first contributions");
        if (!dChunkElemBase) break;
        constexpr int MAXOFF = 1024;
        const int stride = chunkSlots;             // a chunk holds at most chunkSlots rows
        DBuf<int> flag;
        DBuf<int32_t> off;
        flag.alloc(1);
        flag.zero(s);
        off.alloc(MAXOFF);
        dChunkElemBase->alloc((size_t)nChunk);
        hipLaunchKernelGGL(k_sym_chunk_elem_base, dim3((unsigned)nChunk), dim3(64), 0, s, nChunk, dSeg.p, valB.p, dChunkElemBase->p, flag.p, off.p, MAXOFF);
        RP(hipGetLastError());
        int nOff = 0;
        flag.download(&nOff, 1, s);
        lap("  chunks: element span of every chunk");
        if (nOff == 0) { packed = true; break; }
        if (nOff > MAXOFF || round == 3) break;    // no locality to speak of: absolute codes (if they fit)
        std::vector<int32_t> hOff((size_t)nOff);
        off.download(hOff.data(), hOff.size(), s);
        DBuf<uint32_t> rep;
        rep.alloc((size_t)nOff * stride);
        hipLaunchKernelGGL(k_sym_row_rep, dim3(nOff), dim3(256), 0, s, nOff, off.p, dCR.p, rowCStart.p, valB.p, stride, rep.p);
        RP(hipGetLastError());
        std::vector<uint32_t> hRep((size_t)nOff * stride);
        rep.download(hRep.data(), hRep.size(), s);
        for (int o = 0; o < nOff; ++o) {
            const int32_t c = hOff[(size_t)o], r0 = S.chunkRow[c], r1 = S.chunkRow[c + 1];
            uint32_t base = 0xffffffffu;
            for (int32_t r = r0; r < r1; ++r) {
                const uint32_t e = hRep[(size_t)o * stride + (r - r0)];
                if (e == 0xffffffffu) continue;
                if (base == 0xffffffffu) { base = e; continue; }
                const uint32_t d = e > base ? e - base : base - e;
                if (d >= (1u << 24)) { breaks.push_back(r); base = e; }          // half the field: the rows' other elements lie nearby
            }
        }
        std::sort(breaks.begin(), breaks.end());
        breaks.erase(std::unique(breaks.begin(), breaks.end()), breaks.end());
    }
    spmvChunks.join();

    lap("chunks (host): rest");
    // ---- element-major order inside every chunk: sort by (chunk, code)
    DBuf<int32_t> dChunkOfRow, dChunkBase;
    dChunkOfRow.alloc((size_t)nRows);
    dChunkBase.alloc((size_t)nChunk);
    hipLaunchKernelGGL(k_sym_chunk_tables, dim3(grid_of(nChunk)), dim3(256), 0, s, nChunk, dCR.p, dRowPtr.p, dChunkOfRow.p, dChunkBase.p);
    RP(hipGetLastError());
    DBuf<uint16_t> lsA;
    lsA.alloc((size_t)nC);
    if (wantScatter) { dScatter.alloc((size_t)N); MFH_HIP(hipMemsetAsync(dScatter.p, 0xff, (size_t)N * sizeof(int32_t), s)); }
    keyA.release();
    if (!packed && !fitsAbsolute)
        throw Error(MFH_ERR_UNSUPPORTED, "mesh too large for absolute contribution codes and without element locality inside the row chunks (reorder the elements)");
    dContribSlot.alloc((size_t)nC);
    dContribCode.alloc((size_t)nC);
    DBuf<uint32_t> codeA;
    codeA.alloc((size_t)nC);
    hipLaunchKernelGGL(k_sym_key2, dim3(grid_of(nC)), dim3(256), 0, s, nC, keyB.p, valB.p, slotP1.p, dChunkOfRow.p, dChunkBase.p,
                       packed ? dChunkElemBase->p : (const int32_t *)nullptr, cb, (unsigned)(npe * npe), codeA.p, lsA.p, wantScatter ? dScatter.p : nullptr);
    RP(hipGetLastError());
    MFH_HIP(hipStreamSynchronize(s));
    slotP1.release(); rowLen.release(); rowCStart.release(); valB.release(); keyB.release();
    if (codesPacked) *codesPacked = packed;
    // The contributions are already grouped by chunk (chunks are ranges of rows, the first sort ordered the rows): what is left is the
    // order inside each chunk, a SEGMENTED sort of 32-bit codes over ~650-entry segments instead of a second full-length 64-bit sort
    const unsigned endBit2 = packed ? 32u : bits_for((uint64_t)N);
    size_t tmp2 = 0;
    RP(rocprim::segmented_radix_sort_pairs(nullptr, tmp2, codeA.p, dContribCode.p, lsA.p, dContribSlot.p, (unsigned int)nC, (unsigned int)nChunk, dSeg.p,
                                           dSeg.p + 1, 0u, endBit2, s));
    const double tA = now_ms();
    if (tmp2 + 16 > tmp.n) tmp.alloc(tmp2 + 16);
    const double tB = now_ms();
    if (getenv("MFH_SYM_TIMING")) fprintf(stderr, "[symbolic] segmented sort: %.1f MB of temporary storage, allocation %.2f ms\n", tmp2 / 1e6, tB - tA);
    RP(rocprim::segmented_radix_sort_pairs(tmp.p, tmp2, codeA.p, dContribCode.p, lsA.p, dContribSlot.p, (unsigned int)nC, (unsigned int)nChunk, dSeg.p,
                                           dSeg.p + 1, 0u, endBit2, s));
    MFH_HIP(hipStreamSynchronize(s));
    lap("sort by (chunk, code) + lists");
}


// ------------------------------------------------------------------------------------------------
// Gather lists of the matrix-free operator: the nElem*npe (element, local node) pairs grouped by the row
// (DoF of that node), rows grouped into chunks, element-major inside a chunk. Two radix sorts.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) k_mf_gen(int64_t N, const int32_t *__restrict__ elemNodes, const int32_t *__restrict__ dofForNode,
                                                int64_t nRows, uint32_t *__restrict__ key, uint32_t *__restrict__ val,
                                                int32_t *__restrict__ rowCount) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < N; k += (int64_t)gridDim.x * 256) {
        int64_t row = elemNodes[k];
        if (dofForNode) row = dofForNode[row];
        const bool ok = row < nRows;
        key[k] = ok ? (uint32_t)row : (uint32_t)nRows;      // not-owned rows sort last
        val[k] = (uint32_t)k;
        if (ok) atomicAdd(&rowCount[row], 1);
    }
}
__global__ void __launch_bounds__(256) k_mf_key2(int64_t n, const uint32_t *__restrict__ rowKey, const uint32_t *__restrict__ code,
                                                 const int32_t *__restrict__ chunkOfRow, const int32_t *__restrict__ chunkFirstRow,
                                                 uint64_t *__restrict__ key2, uint16_t *__restrict__ lrow) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const int32_t row = (int32_t)rowKey[k];
        const int32_t ch = chunkOfRow[row];
        key2[k] = ((uint64_t)(uint32_t)ch << 32) | (uint64_t)code[k];
        lrow[k] = (uint16_t)(row - chunkFirstRow[ch]);
    }
}
// pos[code] = position of the (element, node) pair in the row-chunk-ordered list (0xffffffff: row not owned)
__global__ void __launch_bounds__(256) k_mf_pos(int64_t n, const uint32_t *__restrict__ code, uint32_t *__restrict__ pos) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) pos[code[k]] = (uint32_t)k;
}
} // namespace

void build_mf_lists_device(const HostMesh &m, const int32_t *dElemNodes, const int32_t *dDofForNode, int64_t nRows, hipStream_t s,
                           MfLists &L, DBuf<uint32_t> &dPairCode, DBuf<uint16_t> &dPairRow, DBuf<uint32_t> &dPairPos, int maxRowsCap,
                           int maxPairs) {
    const int64_t N = m.nElem * m.npe;
    if ((double)N >= 4294967295.0) throw Error(MFH_ERR_UNSUPPORTED, "mesh too large for 32-bit pair codes");
    DBuf<uint32_t> keyA, keyB, valA, valB;
    DBuf<int32_t> rowCount, rowPtr;
    keyA.alloc(N); keyB.alloc(N); valA.alloc(N); valB.alloc(N);
    rowCount.alloc((size_t)nRows + 1);
    rowCount.zero(s);
    hipLaunchKernelGGL(k_mf_gen, dim3(grid_of(N)), dim3(256), 0, s, N, dElemNodes, dDofForNode, nRows, keyA.p, valA.p, rowCount.p);
    RP(hipGetLastError());
    const unsigned endBit1 = bits_for((uint64_t)nRows);
    size_t tmpBytes = 0;
    RP(rocprim::radix_sort_pairs(nullptr, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)N, 0u, endBit1, s));
    DBuf<char> tmp;
    tmp.alloc(tmpBytes + 16);
    RP(rocprim::radix_sort_pairs(tmp.p, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)N, 0u, endBit1, s));
    rowPtr.alloc((size_t)nRows + 1);
    size_t exBytes = 0;
    RP(rocprim::exclusive_scan(nullptr, exBytes, rowCount.p, rowPtr.p, (int32_t)0, (size_t)nRows + 1, rocprim::plus<int32_t>(), s));
    if (exBytes + 16 > tmp.n) tmp.alloc(exBytes + 16);
    RP(rocprim::exclusive_scan(tmp.p, exBytes, rowCount.p, rowPtr.p, (int32_t)0, (size_t)nRows + 1, rocprim::plus<int32_t>(), s));
    std::vector<int32_t> hRowPtr((size_t)nRows + 1);
    rowPtr.download(hRowPtr.data(), hRowPtr.size(), s);
    const int64_t nP = hRowPtr[nRows];
    if (nP == 0) throw Error(MFH_ERR_INVALID, "no element touches an owned row");
    // ---- chunks: <= 256 rows and <= 2048 pairs (a single row may exceed the pair budget: it then forms its own chunk)
    L = MfLists();
    L.chunkRow.push_back(0);
    int64_t r = 0;
    while (r < nRows) {
        int64_t r2 = r + 1;
        while (r2 < nRows && r2 - r < maxRowsCap && hRowPtr[r2 + 1] - hRowPtr[r] <= maxPairs) ++r2;
        L.maxRows = std::max<int>(L.maxRows, (int)(r2 - r));
        L.chunkRow.push_back((int32_t)r2);
        r = r2;
    }
    const int64_t nChunk = (int64_t)L.chunkRow.size() - 1;
    L.pairPtr.resize((size_t)nChunk + 1);
    std::vector<int32_t> chunkOfRow((size_t)nRows), chunkFirst((size_t)nChunk);
    for (int64_t c = 0; c < nChunk; ++c) {
        L.pairPtr[c] = hRowPtr[L.chunkRow[c]];
        chunkFirst[c] = L.chunkRow[c];
        for (int32_t q = L.chunkRow[c]; q < L.chunkRow[c + 1]; ++q) chunkOfRow[q] = (int32_t)c;
    }
    L.pairPtr[nChunk] = nP;
    L.nPairs = nP;
    // ---- element-major order inside every chunk
    DBuf<int32_t> dChunkOfRow, dChunkFirst;
    dChunkOfRow.upload(chunkOfRow, s);
    dChunkFirst.upload(chunkFirst, s);
    DBuf<uint64_t> k2A, k2B;
    DBuf<uint16_t> lrA;
    k2A.alloc((size_t)nP); k2B.alloc((size_t)nP); lrA.alloc((size_t)nP);
    hipLaunchKernelGGL(k_mf_key2, dim3(grid_of(nP)), dim3(256), 0, s, nP, keyB.p, valB.p, dChunkOfRow.p, dChunkFirst.p, k2A.p, lrA.p);
    RP(hipGetLastError());
    keyA.release(); valA.release();
    dPairRow.alloc((size_t)nP);
    const unsigned endBit2 = 32 + bits_for((uint64_t)nChunk);
    size_t tmp2 = 0;
    RP(rocprim::radix_sort_pairs(nullptr, tmp2, k2A.p, k2B.p, lrA.p, dPairRow.p, (size_t)nP, 0u, endBit2, s));
    if (tmp2 + 16 > tmp.n) tmp.alloc(tmp2 + 16);
    RP(rocprim::radix_sort_pairs(tmp.p, tmp2, k2A.p, k2B.p, lrA.p, dPairRow.p, (size_t)nP, 0u, endBit2, s));
    dPairCode.alloc((size_t)nP);
    hipLaunchKernelGGL(k_sym_codes, dim3(grid_of(nP)), dim3(256), 0, s, nP, k2B.p, dPairCode.p);
    RP(hipGetLastError());
    dPairPos.alloc((size_t)N);
    MFH_HIP(hipMemsetAsync(dPairPos.p, 0xff, (size_t)N * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_mf_pos, dim3(grid_of(nP)), dim3(256), 0, s, nP, dPairCode.p, dPairPos.p);
    RP(hipGetLastError());
    MFH_HIP(hipStreamSynchronize(s));
}


// ------------------------------------------------------------------------------------------------
// Lists of the cluster variant of the matrix-free operator (see MfClusterLists). Two radix sorts.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) k_mfc_gen(int64_t N, int npe, int blockElems, const int32_t *__restrict__ blockOfElem,
                                                 const int32_t *__restrict__ elemNodes, const int32_t *__restrict__ dofForNode,
                                                 int64_t nRows, uint64_t *__restrict__ key, uint32_t *__restrict__ val,
                                                 int32_t *__restrict__ rowCount) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < N; k += (int64_t)gridDim.x * 256) {
        int64_t row = elemNodes[k];
        if (dofForNode) row = dofForNode[row];
        const bool ok = row < nRows;
        const uint64_t b = blockOfElem ? (uint64_t)blockOfElem[k / npe] : (uint64_t)((k / npe) / blockElems);
        key[k] = (b << 32) | (uint64_t)row;              // rows of other ranks (row >= nRows) keep their id: x is gathered through the entries
        val[k] = (uint32_t)k;
        if (ok) atomicAdd(&rowCount[row], 1);
    }
}
// per sorted pair: entry id (scan of heads - 1); at heads: entry row, run start; at block heads: blockPtr
__global__ void __launch_bounds__(256) k_mfc_entries(int64_t n, const uint64_t *__restrict__ key, const uint32_t *__restrict__ entP1,
                                                     int32_t *__restrict__ entryRow, int32_t *__restrict__ entryStart,
                                                     int32_t *__restrict__ blockPtr) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const uint64_t kk = key[k];
        if (k != 0 && kk == key[k - 1]) continue;
        const int32_t u = (int32_t)(entP1[k] - 1);
        entryRow[u] = (int32_t)(kk & 0xffffffffu);
        entryStart[u] = (int32_t)k;
        if (k == 0 || (kk >> 32) != (key[k - 1] >> 32)) blockPtr[kk >> 32] = u;
    }
}
// entry classification + interface keys; local index of every pair
__global__ void __launch_bounds__(256) k_mfc_classify(int64_t nU, int64_t nSorted, int64_t nRows, const int32_t *__restrict__ entryRow,
                                                      const int32_t *__restrict__ entryStart, const int32_t *__restrict__ rowCount,
                                                      int32_t *__restrict__ entryDest, int32_t *__restrict__ rowIfaceCount) {
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < nU; u += (int64_t)gridDim.x * 256) {
        const int32_t row = entryRow[u];
        if (row >= nRows) { entryDest[u] = -2; continue; }       // row owned by another rank: a column only
        const int32_t cnt = (u + 1 < nU ? entryStart[u + 1] : (int32_t)nSorted) - entryStart[u];
        if (cnt == rowCount[row]) entryDest[u] = -1;            // every element of the row is in this block
        else { entryDest[u] = 0; atomicAdd(&rowIfaceCount[row], 1); }
    }
}
__global__ void __launch_bounds__(256) k_mfc_local(int64_t n, const uint64_t *__restrict__ key, const uint32_t *__restrict__ val,
                                                   const uint32_t *__restrict__ entP1, const int32_t *__restrict__ blockPtr,
                                                   uint16_t *__restrict__ localIdx) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        localIdx[val[k]] = (uint16_t)((int32_t)(entP1[k] - 1) - blockPtr[key[k] >> 32]);
}
// interface entries: (row, entry) pairs to be sorted by row
__global__ void __launch_bounds__(256) k_mfc_iface_keys(int64_t nU, const int32_t *__restrict__ entryRow, const int32_t *__restrict__ entryDest,
                                                        const uint32_t *__restrict__ ifP1, uint32_t *__restrict__ key, uint32_t *__restrict__ val) {
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < nU; u += (int64_t)gridDim.x * 256)
        if (entryDest[u] == 0) { key[ifP1[u] - 1] = (uint32_t)entryRow[u]; val[ifP1[u] - 1] = (uint32_t)u; }
}
__global__ void __launch_bounds__(256) k_mfc_iface_flags(int64_t nU, const int32_t *__restrict__ entryDest, uint32_t *__restrict__ flag) {
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < nU; u += (int64_t)gridDim.x * 256) flag[u] = entryDest[u] == 0 ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_mfc_iface_assign(int64_t nI, const uint32_t *__restrict__ rowSorted, const uint32_t *__restrict__ entSorted,
                                                          const int32_t *__restrict__ ifaceIdxOfRow, const int32_t *__restrict__ chunkOfIdx,
                                                          const int32_t *__restrict__ chunkFirstIdx, int32_t *__restrict__ entryDest,
                                                          uint16_t *__restrict__ ifaceRow) {
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nI; p += (int64_t)gridDim.x * 256) {
        entryDest[entSorted[p]] = (int32_t)p;
        const int32_t ii = ifaceIdxOfRow[rowSorted[p]];                // index among the interface rows
        ifaceRow[p] = (uint16_t)(ii - chunkFirstIdx[chunkOfIdx[ii]]);
    }
}
__global__ void __launch_bounds__(256) k_mfc_block_of_elem(int64_t nElem, int32_t nBlocks, const int32_t *__restrict__ elemPtr, int32_t *__restrict__ blockOfElem) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nElem; e += (int64_t)gridDim.x * 256) {
        int32_t lo = 0, hi = nBlocks;                     // the block b with elemPtr[b] <= e < elemPtr[b + 1]
        while (hi - lo > 1) {
            const int32_t mid = (lo + hi) >> 1;
            if ((int64_t)elemPtr[mid] <= e) lo = mid; else hi = mid;
        }
        blockOfElem[e] = lo;
    }
}
__global__ void __launch_bounds__(256) k_mfc_rowflag32(int64_t nRows, const int32_t *__restrict__ rowIfaceCount, int32_t *__restrict__ flag) {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r <= nRows; r += (int64_t)gridDim.x * 256) flag[r] = (r < nRows && rowIfaceCount[r] > 0) ? 1 : 0;
}
// interface row ii (compact numbering): its global row, the end of its partials in the row-ordered interface list
__global__ void __launch_bounds__(256) k_mfc_iface_rows(int64_t nRows, const int32_t *__restrict__ rowIfaceCount, const int32_t *__restrict__ idxOfRow,
                                                        const int32_t *__restrict__ rowIPtr, int32_t *__restrict__ rowMap, int32_t *__restrict__ iptr) {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nRows; r += (int64_t)gridDim.x * 256)
        if (rowIfaceCount[r] > 0) { const int32_t ii = idxOfRow[r]; rowMap[ii] = (int32_t)r; iptr[ii + 1] = rowIPtr[r + 1]; }
}
// chunks of the second pass: one thread per group of maxRows consecutive interface rows; chunkBase == null counts, otherwise writes
__global__ void __launch_bounds__(256) k_mfc_chunks(int64_t nGroups, int64_t nIR, int maxRows, int maxPairs, const int32_t *__restrict__ iptr,
                                                    const int32_t *__restrict__ chunkBase, int32_t *__restrict__ chunksOfGroup,
                                                    int32_t *__restrict__ chunkRow, int64_t *__restrict__ pairPtr, int32_t *__restrict__ chunkOfIdx) {
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < nGroups; g += (int64_t)gridDim.x * 256) {
        const int64_t end = min(nIR, (g + 1) * (int64_t)maxRows);
        int32_t cnt = 0;
        for (int64_t r = g * (int64_t)maxRows; r < end;) {
            const int32_t base = iptr[r];
            int64_t r2 = r + 1;
            while (r2 < end && iptr[r2 + 1] - base <= maxPairs) ++r2;
            if (chunkBase) {
                const int32_t ch = chunkBase[g] + cnt;
                chunkRow[ch] = (int32_t)r;
                pairPtr[ch] = base;
                for (int64_t q = r; q < r2; ++q) chunkOfIdx[q] = ch;
            }
            ++cnt;
            r = r2;
        }
        if (!chunkBase) chunksOfGroup[g] = cnt;
    }
}
__global__ void __launch_bounds__(256) k_mfc_rowflag(int64_t nRows, const int32_t *__restrict__ rowIfaceCount, uint8_t *__restrict__ flag) {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < nRows; r += (int64_t)gridDim.x * 256) flag[r] = rowIfaceCount[r] > 0;
}
} // namespace

void build_mf_cluster_lists_device(const HostMesh &m, const int32_t *dElemNodes, const int32_t *dDofForNode, int64_t nRows, hipStream_t s,
                                   MfClusterLists &L, MfClusterDev &D, int blockElems, const std::vector<int32_t> *blockStart) {
    const int npe = m.npe;
    const int64_t N = m.nElem * npe;
    if ((double)N >= 4294967295.0) throw Error(MFH_ERR_UNSUPPORTED, "mesh too large for 32-bit pair codes");
    const bool timing = getenv("MFH_MFC_TIMING") != nullptr;
    double tp = now_ms();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        const double t = now_ms();
        fprintf(stderr, "[operator lists] %-40s %8.2f ms\n", what, t - tp);
        tp = t;
    };
    L = MfClusterLists();
    L.blockElems = std::max(16, std::min(blockElems, MF_BLOCK_ELEMS_MAX));
    L.nBlocks = (m.nElem + L.blockElems - 1) / L.blockElems;
    DBuf<int32_t> dBlockOfElem;
    D.elemPtr.release();
    if (blockStart && blockStart->size() >= 2) {
        L.nBlocks = (int64_t)blockStart->size() - 1;
        D.elemPtr.upload(*blockStart, s);
        dBlockOfElem.alloc((size_t)m.nElem);      // (a host loop over the 40 M elements of a 119^3 grid + the upload of its result took 49 ms)
        hipLaunchKernelGGL(k_mfc_block_of_elem, dim3(grid_of(m.nElem)), dim3(256), 0, s, m.nElem, (int32_t)L.nBlocks, (const int32_t *)D.elemPtr.p, dBlockOfElem.p);
        RP(hipGetLastError());
    }
    lap("block of every element");
    DBuf<uint64_t> keyA, keyB;
    DBuf<uint32_t> valA, valB, entP1;
    DBuf<int32_t> rowCount, rowIfaceCount;
    keyA.alloc(N); keyB.alloc(N); valA.alloc(N); valB.alloc(N);
    rowCount.alloc((size_t)nRows + 1); rowCount.zero(s);
    hipLaunchKernelGGL(k_mfc_gen, dim3(grid_of(N)), dim3(256), 0, s, N, npe, L.blockElems, (const int32_t *)dBlockOfElem.p, dElemNodes, dDofForNode, nRows, keyA.p,
                       valA.p, rowCount.p);
    RP(hipGetLastError());
    const unsigned endBit = 32 + bits_for((uint64_t)L.nBlocks);
    size_t tmpBytes = 0;
    RP(rocprim::radix_sort_pairs(nullptr, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)N, 0u, endBit, s));
    DBuf<char> tmp;
    tmp.alloc(tmpBytes + 16);
    RP(rocprim::radix_sort_pairs(tmp.p, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)N, 0u, endBit, s));
    keyA.release(); valA.release();
    lap("keys + sort by (block, row)");
    // ---- entries = distinct (block, row) pairs
    entP1.alloc(N);
    hipLaunchKernelGGL(k_sym_heads, dim3(grid_of(N)), dim3(256), 0, s, N, keyB.p, entP1.p, 0u);
    size_t scanBytes = 0;
    RP(rocprim::inclusive_scan(nullptr, scanBytes, entP1.p, entP1.p, (size_t)N, rocprim::plus<uint32_t>(), s));
    if (scanBytes + 16 > tmp.n) tmp.alloc(scanBytes + 16);
    RP(rocprim::inclusive_scan(tmp.p, scanBytes, entP1.p, entP1.p, (size_t)N, rocprim::plus<uint32_t>(), s));
    uint32_t nU32 = 0;
    MFH_HIP(hipMemcpyAsync(&nU32, entP1.p + (N - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    const int64_t nU = nU32;
    L.nEntries = nU;
    DBuf<int32_t> entryStart;
    D.entryRow.alloc((size_t)nU); D.entryDest.alloc((size_t)nU); entryStart.alloc((size_t)nU);
    D.blockPtr.alloc((size_t)L.nBlocks + 1);
    hipLaunchKernelGGL(k_mfc_entries, dim3(grid_of(N)), dim3(256), 0, s, N, keyB.p, entP1.p, D.entryRow.p, entryStart.p, D.blockPtr.p);
    RP(hipGetLastError());
    const int32_t nUi = (int32_t)nU;
    MFH_HIP(hipMemcpyAsync(D.blockPtr.p + L.nBlocks, &nUi, sizeof(int32_t), hipMemcpyHostToDevice, s));
    D.localIdx.alloc((size_t)N);
    hipLaunchKernelGGL(k_mfc_local, dim3(grid_of(N)), dim3(256), 0, s, N, keyB.p, valB.p, entP1.p, D.blockPtr.p, D.localIdx.p);
    RP(hipGetLastError());
    std::vector<int32_t> hBlockPtr((size_t)L.nBlocks + 1);
    D.blockPtr.download(hBlockPtr.data(), hBlockPtr.size(), s);
    for (int64_t b = 0; b < L.nBlocks; ++b) L.maxLocal = std::max(L.maxLocal, hBlockPtr[b + 1] - hBlockPtr[b]);
    lap("entries, local indices, block table");
    // ---- classification: finished in the block, or interface
    rowIfaceCount.alloc((size_t)nRows + 1); rowIfaceCount.zero(s);
    hipLaunchKernelGGL(k_mfc_classify, dim3(grid_of(nU)), dim3(256), 0, s, nU, N, nRows, D.entryRow.p, entryStart.p, rowCount.p, D.entryDest.p,
                       rowIfaceCount.p);
    RP(hipGetLastError());
    keyB.release(); valB.release(); entP1.release(); entryStart.release();
    DBuf<uint32_t> ifP1, ikA, ikB, ivA, ivB;
    ifP1.alloc((size_t)nU);
    hipLaunchKernelGGL(k_mfc_iface_flags, dim3(grid_of(nU)), dim3(256), 0, s, nU, D.entryDest.p, ifP1.p);
    RP(rocprim::inclusive_scan(nullptr, scanBytes, ifP1.p, ifP1.p, (size_t)nU, rocprim::plus<uint32_t>(), s));
    if (scanBytes + 16 > tmp.n) tmp.alloc(scanBytes + 16);
    RP(rocprim::inclusive_scan(tmp.p, scanBytes, ifP1.p, ifP1.p, (size_t)nU, rocprim::plus<uint32_t>(), s));
    uint32_t nI32 = 0;
    MFH_HIP(hipMemcpyAsync(&nI32, ifP1.p + (nU - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    const int64_t nI = nI32;
    L.nIface = nI;
    D.rowIsIface.alloc((size_t)nRows);
    hipLaunchKernelGGL(k_mfc_rowflag, dim3(grid_of(nRows)), dim3(256), 0, s, nRows, rowIfaceCount.p, D.rowIsIface.p);
    lap("classification + interface count");
    // ---- second-pass chunks over the rows (only interface entries count)
    DBuf<int32_t> rowIPtr;
    rowIPtr.alloc((size_t)nRows + 1);
    size_t exBytes = 0;
    RP(rocprim::exclusive_scan(nullptr, exBytes, rowIfaceCount.p, rowIPtr.p, (int32_t)0, (size_t)nRows + 1, rocprim::plus<int32_t>(), s));
    if (exBytes + 16 > tmp.n) tmp.alloc(exBytes + 16);
    RP(rocprim::exclusive_scan(tmp.p, exBytes, rowIfaceCount.p, rowIPtr.p, (int32_t)0, (size_t)nRows + 1, rocprim::plus<int32_t>(), s));
    lap("interface row pointers (scan)");
    // The second pass works in the compact numbering of the interface rows (rows with at least one partial). Everything below used to be a host
    // loop over the rows (download of 230 MB of row pointers, 57.6 M iterations, uploads of the tables: 0.24 s at 119^3); it is a scan, a
    // compaction and two small kernels now. Chunks: groups of at most maxRowsCap consecutive interface rows, cut where the partials of a
    // group exceed maxPairs -- one thread walks one group, so the cuts need no sequential pass over all rows.
    const int maxRowsCap = 512, maxPairs = 2048;
    DBuf<int32_t> ifl, dIdxOfRow, iptr, chunksOfGroup, chunkBase, dChunkOfIdx;
    ifl.alloc((size_t)nRows + 1); dIdxOfRow.alloc((size_t)nRows + 1);
    hipLaunchKernelGGL(k_mfc_rowflag32, dim3(grid_of(nRows + 1)), dim3(256), 0, s, nRows, rowIfaceCount.p, ifl.p);
    RP(rocprim::exclusive_scan(nullptr, exBytes, ifl.p, dIdxOfRow.p, (int32_t)0, (size_t)nRows + 1, rocprim::plus<int32_t>(), s));
    if (exBytes + 16 > tmp.n) tmp.alloc(exBytes + 16);
    RP(rocprim::exclusive_scan(tmp.p, exBytes, ifl.p, dIdxOfRow.p, (int32_t)0, (size_t)nRows + 1, rocprim::plus<int32_t>(), s));
    int32_t nIR32 = 0;
    MFH_HIP(hipMemcpyAsync(&nIR32, dIdxOfRow.p + nRows, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    const int64_t nIR = nIR32;
    ifl.release();
    L.nIfaceRows = nIR;
    L.maxRows = (int)std::min<int64_t>(nIR, maxRowsCap);
    D.rowMap.alloc((size_t)std::max<int64_t>(nIR, 1));
    iptr.alloc((size_t)nIR + 1);
    MFH_HIP(hipMemsetAsync(iptr.p, 0, sizeof(int32_t), s));
    if (nIR == 0) MFH_HIP(hipMemsetAsync(D.rowMap.p, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_mfc_iface_rows, dim3(grid_of(nRows)), dim3(256), 0, s, nRows, rowIfaceCount.p, dIdxOfRow.p, rowIPtr.p, D.rowMap.p, iptr.p);
    RP(hipGetLastError());
    const int64_t nGroups = (nIR + maxRowsCap - 1) / maxRowsCap;
    chunksOfGroup.alloc((size_t)nGroups + 1); chunkBase.alloc((size_t)nGroups + 1);
    MFH_HIP(hipMemsetAsync(chunksOfGroup.p + nGroups, 0, sizeof(int32_t), s));
    if (nGroups > 0) {
        hipLaunchKernelGGL(k_mfc_chunks, dim3(grid_of(nGroups)), dim3(256), 0, s, nGroups, nIR, maxRowsCap, maxPairs, iptr.p, (const int32_t *)nullptr,
                           chunksOfGroup.p, (int32_t *)nullptr, (int64_t *)nullptr, (int32_t *)nullptr);
        RP(hipGetLastError());
    }
    RP(rocprim::exclusive_scan(nullptr, exBytes, chunksOfGroup.p, chunkBase.p, (int32_t)0, (size_t)nGroups + 1, rocprim::plus<int32_t>(), s));
    if (exBytes + 16 > tmp.n) tmp.alloc(exBytes + 16);
    RP(rocprim::exclusive_scan(tmp.p, exBytes, chunksOfGroup.p, chunkBase.p, (int32_t)0, (size_t)nGroups + 1, rocprim::plus<int32_t>(), s));
    int32_t nChunk32 = 0;
    MFH_HIP(hipMemcpyAsync(&nChunk32, chunkBase.p + nGroups, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    const int64_t nChunk = nChunk32;
    L.nChunk = nChunk;
    D.chunkRow.alloc((size_t)nChunk + 1);
    D.pairPtr.alloc((size_t)nChunk + 1);
    dChunkOfIdx.alloc((size_t)std::max<int64_t>(nIR, 1));
    if (nGroups > 0) {
        hipLaunchKernelGGL(k_mfc_chunks, dim3(grid_of(nGroups)), dim3(256), 0, s, nGroups, nIR, maxRowsCap, maxPairs, iptr.p, (const int32_t *)chunkBase.p,
                           (int32_t *)nullptr, D.chunkRow.p, D.pairPtr.p, dChunkOfIdx.p);
        RP(hipGetLastError());
    }
    {
        const int32_t endRow = (int32_t)nIR;
        const int64_t endPair = nI;
        MFH_HIP(hipMemcpyAsync(D.chunkRow.p + nChunk, &endRow, sizeof(int32_t), hipMemcpyHostToDevice, s));
        MFH_HIP(hipMemcpyAsync(D.pairPtr.p + nChunk, &endPair, sizeof(int64_t), hipMemcpyHostToDevice, s));
        MFH_HIP(hipStreamSynchronize(s));
    }
    lap("interface rows + chunks");
    D.ifaceRow.alloc((size_t)std::max<int64_t>(nI, 1));
    D.ifaceBuf.alloc((size_t)std::max<int64_t>(nI, 1) * m.dim);
    lap("interface buffers");
    if (nI > 0) {
        ikA.alloc((size_t)nI); ikB.alloc((size_t)nI); ivA.alloc((size_t)nI); ivB.alloc((size_t)nI);
        hipLaunchKernelGGL(k_mfc_iface_keys, dim3(grid_of(nU)), dim3(256), 0, s, nU, D.entryRow.p, D.entryDest.p, ifP1.p, ikA.p, ivA.p);
        size_t t2 = 0;
        const unsigned eb = bits_for((uint64_t)nRows);
        RP(rocprim::radix_sort_pairs(nullptr, t2, ikA.p, ikB.p, ivA.p, ivB.p, (size_t)nI, 0u, eb, s));
        if (t2 + 16 > tmp.n) tmp.alloc(t2 + 16);
        RP(rocprim::radix_sort_pairs(tmp.p, t2, ikA.p, ikB.p, ivA.p, ivB.p, (size_t)nI, 0u, eb, s));
        hipLaunchKernelGGL(k_mfc_iface_assign, dim3(grid_of(nI)), dim3(256), 0, s, nI, ikB.p, ivB.p, dIdxOfRow.p, dChunkOfIdx.p, (const int32_t *)D.chunkRow.p,
                           D.entryDest.p, D.ifaceRow.p);
        RP(hipGetLastError());
        MFH_HIP(hipStreamSynchronize(s));
    }
    lap("interface sort + assignment");
    MFH_HIP(hipStreamSynchronize(s));
}


// ------------------------------------------------------------------------------------------------
// Aggregates of the two-level preconditioner on the device (same bins as build_aggregates, mfh_twolevel.cpp)
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) k_agg_minmax(int64_t n, int dim, const double *__restrict__ pos, double *__restrict__ part) {
    __shared__ double red[6 * 4];
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        for (int a = 0; a < dim; ++a) { const double v = pos[k * dim + a]; mn[a] = fmin(mn[a], v); mx[a] = fmax(mx[a], v); }
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) { mn[a] = fmin(mn[a], __shfl_down(mn[a], off, 64)); mx[a] = fmax(mx[a], __shfl_down(mx[a], off, 64)); }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) for (int a = 0; a < 3; ++a) { red[w * 6 + a] = mn[a]; red[w * 6 + 3 + a] = mx[a]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        part[blockIdx.x * 6 + a] = fmin(fmin(red[a], red[6 + a]), fmin(red[12 + a], red[18 + a]));
        part[blockIdx.x * 6 + 3 + a] = fmax(fmax(red[3 + a], red[9 + a]), fmax(red[15 + a], red[21 + a]));
    }
}
struct AggBins { double mn[3], w[3]; int nb[3]; };
__global__ void __launch_bounds__(256) k_agg_rawbin(int64_t n, int dim, AggBins B, const double *__restrict__ pos, int32_t *__restrict__ rawBin,
                                                    int32_t *__restrict__ mark) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        int ib[3] = {0, 0, 0};
        for (int a = 0; a < dim; ++a) {
            int b = B.w[a] > 0 ? (int)floor((pos[k * dim + a] - B.mn[a]) / B.w[a]) : 0;
            ib[a] = min(max(b, 0), B.nb[a] - 1);
        }
        const int32_t r = (ib[2] * B.nb[1] + ib[1]) * B.nb[0] + ib[0];
        rawBin[k] = r;
        mark[r] = 1;
    }
}
__global__ void __launch_bounds__(256) k_agg_assign(int64_t n, const int32_t *__restrict__ rawBin, const int32_t *__restrict__ binId,
                                                    int32_t *__restrict__ aggOfDof, uint32_t *__restrict__ key, uint32_t *__restrict__ val,
                                                    int32_t *__restrict__ cnt) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const int32_t a = binId[rawBin[k]];
        aggOfDof[k] = a;
        key[k] = (uint32_t)a;
        val[k] = (uint32_t)k;
        atomicAdd(&cnt[a], 1);
    }
}
// one workgroup per aggregate: centroid = mean position of its DoFs
__global__ void __launch_bounds__(256) k_agg_centroid(int dim, const int32_t *__restrict__ aggPtr, const uint32_t *__restrict__ dofsByAgg,
                                                      const double *__restrict__ pos, double *__restrict__ centroid) {
    __shared__ double red[3 * 4];
    const int a = blockIdx.x;
    double acc[3] = {0, 0, 0};
    for (int q = aggPtr[a] + threadIdx.x; q < aggPtr[a + 1]; q += 256) {
        const int64_t n = dofsByAgg[q];
        for (int c = 0; c < dim; ++c) acc[c] += pos[n * dim + c];
    }
    for (int c = 0; c < 3; ++c)
        for (int off = 32; off > 0; off >>= 1) acc[c] += __shfl_down(acc[c], off, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) for (int c = 0; c < 3; ++c) red[w * 3 + c] = acc[c];
    __syncthreads();
    if (threadIdx.x < 3) {
        const int cnt = aggPtr[a + 1] - aggPtr[a];
        centroid[a * 3 + threadIdx.x] = (red[threadIdx.x] + red[3 + threadIdx.x] + red[6 + threadIdx.x] + red[9 + threadIdx.x]) / (cnt > 0 ? cnt : 1);
    }
}
__global__ void __launch_bounds__(256) k_agg_relpos(int64_t n, int dim, double invH, const int32_t *__restrict__ aggOfDof,
                                                    const double *__restrict__ pos, const double *__restrict__ centroid, double *__restrict__ relPos) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const int a = aggOfDof[k];
        for (int c = 0; c < 3; ++c) relPos[k * 3 + c] = c < dim ? (pos[k * dim + c] - centroid[a * 3 + c]) * invH : 0.0;
    }
}
__global__ void __launch_bounds__(256) k_agg_copy_u32_i32(int64_t n, const uint32_t *__restrict__ a, int32_t *__restrict__ b) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) b[k] = (int32_t)a[k];
}
} // namespace

namespace {
__global__ void __launch_bounds__(256) k_dof_first_node(int64_t nNode, const int32_t *__restrict__ dofForNode, int32_t *__restrict__ first) {
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nNode; n += (int64_t)gridDim.x * 256) atomicMin(&first[dofForNode[n]], (int32_t)n);
}
__global__ void __launch_bounds__(256) k_dof_pos(int64_t nDoF, int dim, const int32_t *__restrict__ first, const double *__restrict__ nodePos,
                                                 double *__restrict__ dofPos) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nDoF; q += (int64_t)gridDim.x * 256)
        for (int a = 0; a < dim; ++a) dofPos[q * dim + a] = nodePos[(int64_t)first[q] * dim + a];
}
} // namespace

namespace {
__global__ void __launch_bounds__(256) k_wrap_positions(int64_t n, int dim, double lx, double ly, double lz, double hx, double hy, double hz, double eps,
                                                        int skipDims, double *__restrict__ pos) {
    const double lo[3] = {lx, ly, lz}, hi[3] = {hx, hy, hz};
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n; q += (int64_t)gridDim.x * 256)
        for (int a = 0; a < dim; ++a)
            if (!((skipDims >> a) & 1) && pos[q * dim + a] >= hi[a] - eps) pos[q * dim + a] = lo[a];
}
} // namespace
// positions on the maximal faces of the box go to the minimal ones: every periodic image of a point then sits at the same place, whichever image
// a rank happens to hold (aggregates of a periodic cell on a partitioned context)
// (skipDims: bit a set = axis a is not periodic -- option "periodic_ignore_dims" -- and keeps its maximal face)
void wrap_positions_device(int64_t n, int dim, const double box[6], hipStream_t s, double *dPos, int skipDims) {
    double ext = 0;
    for (int a = 0; a < dim; ++a) ext = std::max(ext, box[3 + a] - box[a]);
    hipLaunchKernelGGL(k_wrap_positions, dim3(grid_of(n)), dim3(256), 0, s, n, dim, box[0], box[1], box[2], box[3], box[4], box[5], 1e-9 * ext, skipDims, dPos);
    RP(hipGetLastError());
}
// position of a DoF = position of its first node (periodic DoF maps)
void dof_positions_device(int64_t nNode, int dim, const int32_t *dDofForNode, const double *dNodePos, int64_t nDoF, hipStream_t s,
                          DBuf<double> &out) {
    DBuf<int32_t> first;
    first.alloc((size_t)nDoF);
    MFH_HIP(hipMemsetAsync(first.p, 0x7f, (size_t)nDoF * sizeof(int32_t), s));
    hipLaunchKernelGGL(k_dof_first_node, dim3(grid_of(nNode)), dim3(256), 0, s, nNode, dDofForNode, first.p);
    out.alloc((size_t)nDoF * dim);
    hipLaunchKernelGGL(k_dof_pos, dim3(grid_of(nDoF)), dim3(256), 0, s, nDoF, dim, first.p, dNodePos, out.p);
    RP(hipGetLastError());
    MFH_HIP(hipStreamSynchronize(s));
}

// ------------------------------------------------------------------------------------------------
// Transfer lists of the p-multigrid preconditioner (mfh_multigrid.cpp) for a mesh in the library's own numbering (vertices are
// the nodes [0, nVert), identity DoF map): the two ends of every edge node from the element table, and the CSR lists
// vertex -> edge nodes it is an end of (one stable radix sort of the 2 (nNode - nVert) (end, edge node) pairs).
// ------------------------------------------------------------------------------------------------
namespace {
__constant__ int kMgEdgeS[6] = {0, 1, 2, 0, 2, 1};      // Simplex.hh:43-44
__constant__ int kMgEdgeE[6] = {1, 2, 0, 3, 3, 3};
__global__ void __launch_bounds__(256) k_mg_parents(int64_t nElem, int npe, int nv, int dim, const int32_t *__restrict__ elemNodes, int64_t nVert,
                                                    int32_t *__restrict__ parA, int32_t *__restrict__ parB) {
    const int nedge = npe - nv;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nElem * nedge; k += (int64_t)gridDim.x * 256) {
        const int64_t e = k / nedge;
        const int ei = (int)(k - e * nedge);
        const int32_t *en = elemNodes + e * npe;
        int32_t a = en[dim == 3 ? kMgEdgeS[ei] : ei], b = en[dim == 3 ? kMgEdgeE[ei] : (ei + 1) % 3];
        if (a > b) { const int32_t t = a; a = b; b = t; }
        const int32_t f = en[nv + ei];
        parA[f] = a; parB[f] = b;                 // every element of the edge stores the same pair
    }
}
__global__ void __launch_bounds__(256) k_mg_vertex_identity(int64_t nVert, int32_t *__restrict__ parA, int32_t *__restrict__ parB, int32_t *__restrict__ fineOf) {
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nVert; v += (int64_t)gridDim.x * 256) { parA[v] = parB[v] = (int32_t)v; fineOf[v] = (int32_t)v; }
}
__global__ void __launch_bounds__(256) k_mg_pairs(int64_t nEdgeNodes, int64_t nVert, const int32_t *__restrict__ parA, const int32_t *__restrict__ parB,
                                                  uint32_t *__restrict__ key, uint32_t *__restrict__ val, int32_t *__restrict__ cnt) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nEdgeNodes; k += (int64_t)gridDim.x * 256) {
        const int32_t f = (int32_t)(nVert + k);
        key[2 * k] = (uint32_t)parA[f]; val[2 * k] = (uint32_t)f;
        key[2 * k + 1] = (uint32_t)parB[f]; val[2 * k + 1] = (uint32_t)f;
        atomicAdd(&cnt[parA[f]], 1);
        atomicAdd(&cnt[parB[f]], 1);
    }
}
} // namespace

void build_mg_transfer_device(const HostMesh &m, const int32_t *dElemNodes, hipStream_t s, DBuf<int32_t> &parA, DBuf<int32_t> &parB,
                              DBuf<int32_t> &fineOf, DBuf<int32_t> &resPtr, DBuf<int32_t> &resIdx) {
    const int64_t nVert = m.nVert, nEdgeNodes = m.nNode - m.nVert;
    const int nv = m.dim + 1;
    parA.alloc((size_t)m.nNode); parB.alloc((size_t)m.nNode); fineOf.alloc((size_t)nVert);
    hipLaunchKernelGGL(k_mg_vertex_identity, dim3(grid_of(nVert)), dim3(256), 0, s, nVert, parA.p, parB.p, fineOf.p);
    hipLaunchKernelGGL(k_mg_parents, dim3(grid_of(m.nElem * (m.npe - nv))), dim3(256), 0, s, m.nElem, m.npe, nv, m.dim, dElemNodes, nVert, parA.p, parB.p);
    RP(hipGetLastError());
    const int64_t N = 2 * nEdgeNodes;
    DBuf<uint32_t> keyA, keyB, valA, valB;
    DBuf<int32_t> cnt;
    keyA.alloc((size_t)std::max<int64_t>(1, N)); keyB.alloc(keyA.n); valA.alloc(keyA.n); valB.alloc(keyA.n);
    cnt.alloc((size_t)nVert + 1);
    cnt.zero(s);
    hipLaunchKernelGGL(k_mg_pairs, dim3(grid_of(std::max<int64_t>(1, nEdgeNodes))), dim3(256), 0, s, nEdgeNodes, nVert, parA.p, parB.p, keyA.p, valA.p, cnt.p);
    RP(hipGetLastError());
    size_t tmpBytes = 0;
    const unsigned endBit = bits_for((uint64_t)nVert);
    RP(rocprim::radix_sort_pairs(nullptr, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)N, 0u, endBit, s));
    DBuf<char> tmp;
    tmp.alloc(tmpBytes + 16);
    RP(rocprim::radix_sort_pairs(tmp.p, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)N, 0u, endBit, s));
    resPtr.alloc((size_t)nVert + 1);
    size_t exBytes = 0;
    RP(rocprim::exclusive_scan(nullptr, exBytes, cnt.p, resPtr.p, (int32_t)0, (size_t)nVert + 1, rocprim::plus<int32_t>(), s));
    if (exBytes + 16 > tmp.n) tmp.alloc(exBytes + 16);
    RP(rocprim::exclusive_scan(tmp.p, exBytes, cnt.p, resPtr.p, (int32_t)0, (size_t)nVert + 1, rocprim::plus<int32_t>(), s));
    resIdx.alloc((size_t)std::max<int64_t>(1, N));
    hipLaunchKernelGGL(k_agg_copy_u32_i32, dim3(grid_of(std::max<int64_t>(1, N))), dim3(256), 0, s, N, valB.p, resIdx.p);
    RP(hipGetLastError());
    MFH_HIP(hipStreamSynchronize(s));
}

namespace {
// per-workgroup sums of the elements' extents along every axis (max - min over the element's corners)
__global__ void __launch_bounds__(256) k_elem_extent(int64_t nElem, int dim, int npe, const int32_t *__restrict__ elemNodes, const double *__restrict__ pos,
                                                     double *__restrict__ part) {
    __shared__ double red[3 * 256];
    double acc[3] = {0, 0, 0};
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nElem; e += (int64_t)gridDim.x * 256) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int k = 0; k <= dim; ++k) {
            const int64_t v = elemNodes[e * npe + k];
            for (int a = 0; a < dim; ++a) { const double x = pos[v * dim + a]; lo[a] = x < lo[a] ? x : lo[a]; hi[a] = x > hi[a] ? x : hi[a]; }
        }
        for (int a = 0; a < dim; ++a) acc[a] += hi[a] - lo[a];
    }
    for (int a = 0; a < 3; ++a) red[a * 256 + threadIdx.x] = acc[a];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) for (int a = 0; a < 3; ++a) red[a * 256 + threadIdx.x] += red[a * 256 + threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 3) part[blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x * 256];
}
}   // namespace
// sum over the elements of their extent along every axis (the caller divides by the element count: the mean element size per axis)
void element_extent_sums_device(int dim, int64_t nElem, int npe, const int32_t *dElemNodes, const double *dPos, hipStream_t s, double out[3]) {
    const int grid = 512;
    DBuf<double> part;
    part.alloc((size_t)grid * 3);
    hipLaunchKernelGGL(k_elem_extent, dim3(grid), dim3(256), 0, s, nElem, dim, npe, dElemNodes, dPos, part.p);
    RP(hipGetLastError());
    std::vector<double> hp((size_t)grid * 3);
    part.download(hp.data(), hp.size(), s);
    for (int a = 0; a < 3; ++a) out[a] = 0.0;
    for (int b = 0; b < grid; ++b) for (int a = 0; a < dim; ++a) out[a] += hp[(size_t)b * 3 + a];
}

void bounding_box_device(int dim, int64_t nDoF, const double *dPos, hipStream_t s, double mn[3], double mx[3]) {
    const int gridMM = 1024;
    DBuf<double> part;
    part.alloc((size_t)gridMM * 6);
    hipLaunchKernelGGL(k_agg_minmax, dim3(gridMM), dim3(256), 0, s, nDoF, dim, dPos, part.p);
    RP(hipGetLastError());
    std::vector<double> hp((size_t)gridMM * 6);
    part.download(hp.data(), hp.size(), s);
    for (int a = 0; a < 3; ++a) { mn[a] = 1e300; mx[a] = -1e300; }
    for (int b = 0; b < gridMM; ++b)
        for (int a = 0; a < dim; ++a) { mn[a] = std::min(mn[a], hp[(size_t)b * 6 + a]); mx[a] = std::max(mx[a], hp[(size_t)b * 6 + 3 + a]); }
}

// globalBox (mn[3], mx[3]) / globalCount: the lattice of a mesh this context holds a part of (row-partitioned contexts: every rank
// lays the same bins over the whole mesh). fullLattice: every bin is an aggregate, empty or not, numbered in bin order, and the
// reference point of its rigid-body modes is the bin centre -- the same numbers on every rank without any exchange.
void build_aggregates_device(int dim, int64_t nDoF, const double *dPos, int targetNodes, hipStream_t s, Aggregates &A,
                             DBuf<int32_t> &dAggOfDof, DBuf<double> &dRelPos, DBuf<int32_t> &dAggPtr, DBuf<int32_t> &dDofsByAgg,
                             const double *globalBox, int64_t globalCount, bool fullLattice, const double *aspect) {
    A = Aggregates();
    A.dim = dim;
    // ---- bounding box
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    if (globalBox) { for (int a = 0; a < 3; ++a) { mn[a] = globalBox[a]; mx[a] = globalBox[3 + a]; } }
    else bounding_box_device(dim, nDoF, dPos, s, mn, mx);
    // ---- the same bin lattice as build_aggregates (mfh_twolevel.cpp)
    double vol = 1;
    for (int a = 0; a < dim; ++a) vol *= std::max(mx[a] - mn[a], 1e-300);
    const double H = std::pow(vol * std::max(1, targetNodes) / (double)std::max<int64_t>(1, globalCount > 0 ? globalCount : nDoF), 1.0 / dim);
    // aspect (may be null): relative element size per axis. The bins then have the proportions of the elements -- about the same number
    // of elements across a bin along every axis -- at the same bin volume H^dim; the rigid-body modes keep the one length scale H.
    double Ha[3] = {H, H, H};
    if (aspect) {
        double g = 1;
        for (int a = 0; a < dim; ++a) g *= aspect[a];
        g = std::pow(g, 1.0 / dim);
        for (int a = 0; a < dim; ++a) Ha[a] = H * aspect[a] / g;
    }
    int nb[3] = {1, 1, 1};
    for (int a = 0; a < dim; ++a) nb[a] = std::max(1, (int)std::floor((mx[a] - mn[a]) / Ha[a] + 0.5));
    A.H = H;
    AggBins B{};
    for (int a = 0; a < 3; ++a) { B.mn[a] = a < dim ? mn[a] : 0.0; B.nb[a] = nb[a]; B.w[a] = a < dim ? (mx[a] - mn[a]) / nb[a] : 0.0; }
    const int64_t nBins = (int64_t)nb[0] * nb[1] * nb[2];
    DBuf<int32_t> rawBin, mark, dBinId, cnt;
    rawBin.alloc((size_t)nDoF);
    mark.alloc((size_t)nBins);
    mark.zero(s);
    hipLaunchKernelGGL(k_agg_rawbin, dim3(grid_of(nDoF)), dim3(256), 0, s, nDoF, dim, B, dPos, rawBin.p, mark.p);
    RP(hipGetLastError());
    std::vector<int32_t> binId((size_t)nBins);
    mark.download(binId.data(), binId.size(), s);
    int32_t nAgg = 0;
    if (fullLattice) {
        if (nBins > 2000000000LL) throw Error(MFH_ERR_UNSUPPORTED, "too many lattice bins");
        for (auto &b : binId) b = nAgg++;
    } else
        for (auto &b : binId) b = b ? nAgg++ : -1;      // compact numbering of the non-empty bins, in bin order
    A.nAgg = nAgg;
    dBinId.upload(binId, s);
    // ---- aggregate of every DoF, DoFs grouped by aggregate (stable sort: ascending DoF inside an aggregate)
    DBuf<uint32_t> keyA, keyB, valA, valB;
    keyA.alloc((size_t)nDoF); keyB.alloc((size_t)nDoF); valA.alloc((size_t)nDoF); valB.alloc((size_t)nDoF);
    cnt.alloc((size_t)nAgg + 1);
    cnt.zero(s);
    dAggOfDof.alloc((size_t)nDoF);
    hipLaunchKernelGGL(k_agg_assign, dim3(grid_of(nDoF)), dim3(256), 0, s, nDoF, rawBin.p, dBinId.p, dAggOfDof.p, keyA.p, valA.p, cnt.p);
    RP(hipGetLastError());
    size_t tmpBytes = 0;
    const unsigned endBit = bits_for((uint64_t)nAgg);
    RP(rocprim::radix_sort_pairs(nullptr, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)nDoF, 0u, endBit, s));
    DBuf<char> tmp;
    tmp.alloc(tmpBytes + 16);
    RP(rocprim::radix_sort_pairs(tmp.p, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)nDoF, 0u, endBit, s));
    std::vector<int32_t> hCnt((size_t)nAgg + 1);
    cnt.download(hCnt.data(), hCnt.size(), s);
    A.aggPtr.assign((size_t)nAgg + 1, 0);
    for (int32_t a = 0; a < nAgg; ++a) A.aggPtr[(size_t)a + 1] = A.aggPtr[a] + hCnt[a];
    dAggPtr.upload(A.aggPtr, s);
    dDofsByAgg.alloc((size_t)nDoF);
    hipLaunchKernelGGL(k_agg_copy_u32_i32, dim3(grid_of(nDoF)), dim3(256), 0, s, nDoF, valB.p, dDofsByAgg.p);
    // ---- centroids and relative positions
    DBuf<double> dCentroid;
    dCentroid.alloc((size_t)nAgg * 3);
    if (fullLattice) {
        std::vector<double> centre((size_t)nAgg * 3, 0.0);
        for (int iz = 0; iz < nb[2]; ++iz)
            for (int iy = 0; iy < nb[1]; ++iy)
                for (int ix = 0; ix < nb[0]; ++ix) {
                    const size_t b = ((size_t)iz * nb[1] + iy) * nb[0] + ix;
                    const int q[3] = {ix, iy, iz};
                    for (int a = 0; a < dim; ++a) centre[b * 3 + a] = B.mn[a] + (q[a] + 0.5) * B.w[a];
                }
        dCentroid.upload(centre, s);
    } else
        hipLaunchKernelGGL(k_agg_centroid, dim3(nAgg), dim3(256), 0, s, dim, dAggPtr.p, valB.p, dPos, dCentroid.p);
    dRelPos.alloc((size_t)nDoF * 3);
    hipLaunchKernelGGL(k_agg_relpos, dim3(grid_of(nDoF)), dim3(256), 0, s, nDoF, dim, 1.0 / H, dAggOfDof.p, dPos, dCentroid.p, dRelPos.p);
    RP(hipGetLastError());
    A.centroid.resize((size_t)nAgg * 3);
    dCentroid.download(A.centroid.data(), A.centroid.size(), s);
    aggregate_lattice_tables(dim, nb, binId, A);
}


// ------------------------------------------------------------------------------------------------
// FEM mesh topology on the device: P2 edge-node numbering in first-encounter order (FEMMesh.inl:22-36)
// and the unmatched half-faces / half-edges that form the boundary (TetMesh.inl:36-79,
// TriMesh.inl:60-100), by radix sorts instead of the host's hash table + std::sort.
// ------------------------------------------------------------------------------------------------
namespace {
__constant__ int kEdgeS[6] = {0, 1, 2, 0, 2, 1};
__constant__ int kEdgeE[6] = {1, 2, 0, 3, 3, 3};
__constant__ int kFaceC[4][3] = {{1, 3, 2}, {0, 2, 3}, {0, 3, 1}, {0, 1, 2}};

__global__ void __launch_bounds__(256) k_topo_edge_keys(int64_t nInst, int nv, int nedge, const int32_t *__restrict__ ev,
                                                        uint64_t *__restrict__ key, uint32_t *__restrict__ val) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nInst; k += (int64_t)gridDim.x * 256) {
        const int64_t e = k / nedge;
        const int ei = (int)(k - e * nedge);
        const uint32_t a = (uint32_t)ev[e * nv + kEdgeS[ei]], b = (uint32_t)ev[e * nv + kEdgeE[ei]];
        key[k] = a < b ? (((uint64_t)a << 32) | b) : (((uint64_t)b << 32) | a);
        val[k] = (uint32_t)k;
    }
}
// the node table and the node positions of the FEMMesh, written where the assembly reads them (the host builds its own copies for the
// queries of the API meanwhile): corners, then for quadratic elements the edge nodes nVert + instEdge and their midpoints -- every
// element of an edge stores the same 0.5 (a + b), bit for bit what compute_node_positions stores on the host
__global__ void __launch_bounds__(256) k_topo_node_tables(int64_t nElem, int dim, int deg, int64_t nVert, const int32_t *__restrict__ ev,
                                                          const int32_t *__restrict__ instEdge, int32_t *__restrict__ elemNodes,
                                                          double *__restrict__ nodePos) {
    const int nv = dim + 1, nedge = deg == 2 ? (dim == 3 ? 6 : 3) : 0, npe = nv + nedge;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nElem; e += (int64_t)gridDim.x * 256) {
        int32_t v[4];
        for (int c = 0; c < nv; ++c) { v[c] = ev[e * nv + c]; elemNodes[e * npe + c] = v[c]; }
        for (int ei = 0; ei < nedge; ++ei) {
            const int64_t node = nVert + instEdge[e * nedge + ei];
            elemNodes[e * npe + nv + ei] = (int32_t)node;
            const double *pa = nodePos + (int64_t)v[kEdgeS[ei]] * dim, *pb = nodePos + (int64_t)v[kEdgeE[ei]] * dim;
            for (int a = 0; a < dim; ++a) nodePos[node * dim + a] = 0.5 * (pa[a] + pb[a]);
        }
    }
}
// unique edges: at heads, uniq id = headCount-1; record first instance (smallest k: the sort is stable)
__global__ void __launch_bounds__(256) k_topo_edge_first(int64_t n, const uint64_t *__restrict__ key, const uint32_t *__restrict__ val,
                                                         const uint32_t *__restrict__ headP1, uint32_t *__restrict__ firstInst) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        if (k == 0 || key[k] != key[k - 1]) firstInst[headP1[k] - 1] = val[k];
}
__global__ void __launch_bounds__(256) k_topo_iota(int64_t n, uint32_t *__restrict__ v) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) v[k] = (uint32_t)k;
}
// rankOfUniq[uniqSortedByFirst[r]] = r
__global__ void __launch_bounds__(256) k_topo_rank(int64_t n, const uint32_t *__restrict__ uniqByFirst, uint32_t *__restrict__ rankOfUniq) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) rankOfUniq[uniqByFirst[k]] = (uint32_t)k;
}
// edge node of every (element, local edge) instance
__global__ void __launch_bounds__(256) k_topo_edge_assign(int64_t n, const uint32_t *__restrict__ val, const uint32_t *__restrict__ headP1,
                                                          const uint32_t *__restrict__ rankOfUniq, int32_t *__restrict__ instEdge) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) instEdge[val[k]] = (int32_t)rankOfUniq[headP1[k] - 1];
}
// half-face keys (3 x 21-bit sorted vertex ids) / half-edge keys
__global__ void __launch_bounds__(256) k_topo_face_keys(int64_t nInst, int dim, const int32_t *__restrict__ ev, uint64_t *__restrict__ key,
                                                        uint32_t *__restrict__ val) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nInst; k += (int64_t)gridDim.x * 256) {
        if (dim == 3) {
            const int64_t t = k >> 2;
            const int f = (int)(k & 3);
            const uint64_t x = (uint32_t)ev[t * 4 + kFaceC[f][0]], y = (uint32_t)ev[t * 4 + kFaceC[f][1]], z = (uint32_t)ev[t * 4 + kFaceC[f][2]];
            const uint64_t lo = min(x, min(y, z)), hi = max(x, max(y, z)), mid = x ^ y ^ z ^ lo ^ hi;
            key[k] = (lo << 42) | (mid << 21) | hi;
        } else {
            const int64_t t = k / 3;
            const int c = (int)(k - t * 3);
            const uint64_t tail = (uint32_t)ev[t * 3 + (c + 1) % 3], tip = (uint32_t)ev[t * 3 + (c + 2) % 3];
            key[k] = (min(tail, tip) << 32) | max(tail, tip);
        }
        val[k] = (uint32_t)k;
    }
}
// flag: 1 = unmatched (boundary), 2 = more than two incident (non-manifold)
__global__ void __launch_bounds__(256) k_topo_face_flags(int64_t n, const uint64_t *__restrict__ key, uint32_t *__restrict__ flag, int *nonManifold) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const bool sameP = k > 0 && key[k] == key[k - 1], sameN = k + 1 < n && key[k] == key[k + 1];
        flag[k] = (!sameP && !sameN) ? 1u : 0u;
        if (sameP && sameN) *nonManifold = 1;
    }
}
// wide variant for meshes with >= 2^21 vertices: the sorted (lo, mid, hi) vertex triple does not fit one 64-bit key, so the
// faces are ordered by two stable sorts (hi, then lo:mid) and the triples are compared through the instance ids
__device__ inline void topo_face_triple(const int32_t *__restrict__ ev, uint32_t inst, uint32_t &lo, uint32_t &mid, uint32_t &hi) {
    const int64_t t = inst >> 2;
    const int f = (int)(inst & 3);
    const uint32_t x = (uint32_t)ev[t * 4 + kFaceC[f][0]], y = (uint32_t)ev[t * 4 + kFaceC[f][1]], z = (uint32_t)ev[t * 4 + kFaceC[f][2]];
    lo = min(x, min(y, z)); hi = max(x, max(y, z)); mid = x ^ y ^ z ^ lo ^ hi;
}
__global__ void __launch_bounds__(256) k_topo_face_keys_wide(int64_t nInst, const int32_t *__restrict__ ev, uint32_t *__restrict__ keyHi32,
                                                             uint32_t *__restrict__ val) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < nInst; k += (int64_t)gridDim.x * 256) {
        uint32_t lo, mid, hi;
        topo_face_triple(ev, (uint32_t)k, lo, mid, hi);
        keyHi32[k] = hi;
        val[k] = (uint32_t)k;
    }
}
__global__ void __launch_bounds__(256) k_topo_face_keys_lomid(int64_t n, const int32_t *__restrict__ ev, const uint32_t *__restrict__ inst,
                                                              uint64_t *__restrict__ key) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        uint32_t lo, mid, hi;
        topo_face_triple(ev, inst[k], lo, mid, hi);
        key[k] = ((uint64_t)lo << 32) | mid;
    }
}
__global__ void __launch_bounds__(256) k_topo_face_flags_wide(int64_t n, const int32_t *__restrict__ ev, const uint32_t *__restrict__ inst,
                                                              uint32_t *__restrict__ flag, int *nonManifold) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        uint32_t a0, a1, a2, b0, b1, b2;
        topo_face_triple(ev, inst[k], a0, a1, a2);
        bool sameP = false, sameN = false;
        if (k > 0) { topo_face_triple(ev, inst[k - 1], b0, b1, b2); sameP = a0 == b0 && a1 == b1 && a2 == b2; }
        if (k + 1 < n) { topo_face_triple(ev, inst[k + 1], b0, b1, b2); sameN = a0 == b0 && a1 == b1 && a2 == b2; }
        flag[k] = (!sameP && !sameN) ? 1u : 0u;
        if (sameP && sameN) *nonManifold = 1;
    }
}
__global__ void __launch_bounds__(256) k_topo_face_compact(int64_t n, const uint32_t *__restrict__ val, const uint32_t *__restrict__ flag,
                                                           const uint32_t *__restrict__ posP1, uint32_t *__restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        if (flag[k]) out[posP1[k] - 1] = val[k];
}
} // namespace

// Always returns true (kept for the caller's host fallback switch); meshes with >= 2^21 vertices in 3D use the
// two-sort variant of the half-face ordering. instEdge: nElem*nedge first-encounter edge-node ranks (deg 2 only);
// bdryInst: half-face (4t+f) / half-edge (3t+c) instances without a mate, in sorted-key order.
bool build_topology_device(int dim, int deg, int64_t nElem, int64_t nVert, const int32_t *hostElemVerts, hipStream_t s,
                           RawVec<int32_t> &instEdge, int32_t &nEdgeNodes, std::vector<uint32_t> &bdryInst,
                           const double *hostVertPos, DBuf<int32_t> *dElemNodesOut, DBuf<double> *dNodePosOut,
                           const std::function<void(int, int32_t)> &hostOverlap) {
    // packed 3 x 21-bit face keys do not fit beyond 2^21 vertices: two-sort variant below (MFH_TOPO_FORCE_WIDE: tests)
    const bool wide = dim == 3 && (nVert >= (1 << 21) || getenv("MFH_TOPO_FORCE_WIDE") != nullptr);
    const int nv = dim + 1, nedge = dim == 3 ? 6 : 3;
    DBuf<int32_t> dEv;
    dEv.upload(hostElemVerts, (size_t)nElem * nv, s);
    DBuf<char> tmp;
    auto ensureTmp = [&](size_t bytes) { if (bytes + 16 > tmp.n) tmp.alloc(bytes + 16); };
    nEdgeNodes = 0;
    instEdge.clear();
    DBuf<int32_t> dInst;
    if (deg == 2) {
        const int64_t n = nElem * nedge;
        DBuf<uint64_t> kA, kB;
        DBuf<uint32_t> vA, vB, headP1;
        kA.alloc(n); kB.alloc(n); vA.alloc(n); vB.alloc(n); headP1.alloc(n);
        hipLaunchKernelGGL(k_topo_edge_keys, dim3(grid_of(n)), dim3(256), 0, s, n, nv, nedge, dEv.p, kA.p, vA.p);
        size_t b = 0;
        RP(rocprim::radix_sort_pairs(nullptr, b, kA.p, kB.p, vA.p, vB.p, (size_t)n, 0u, 32 + bits_for((uint64_t)nVert), s));
        ensureTmp(b);
        RP(rocprim::radix_sort_pairs(tmp.p, b, kA.p, kB.p, vA.p, vB.p, (size_t)n, 0u, 32 + bits_for((uint64_t)nVert), s));
        // the host table the ranks are downloaded into (0.97 GB at 119^3) takes its page faults on all host threads while the sort runs,
        // and so does whatever the caller wants sized meanwhile (stage 0: the element node table)
        resize_prefaulted(instEdge, (size_t)n);
        if (hostOverlap) hostOverlap(0, 0);
        hipLaunchKernelGGL(k_sym_heads, dim3(grid_of(n)), dim3(256), 0, s, n, kB.p, headP1.p, 0u);
        RP(rocprim::inclusive_scan(nullptr, b, headP1.p, headP1.p, (size_t)n, rocprim::plus<uint32_t>(), s));
        ensureTmp(b);
        RP(rocprim::inclusive_scan(tmp.p, b, headP1.p, headP1.p, (size_t)n, rocprim::plus<uint32_t>(), s));
        uint32_t nU = 0;
        MFH_HIP(hipMemcpyAsync(&nU, headP1.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        MFH_HIP(hipStreamSynchronize(s));
        nEdgeNodes = (int32_t)nU;
        // order the unique edges by their first instance = first-encounter numbering
        DBuf<uint32_t> firstInst, firstSorted, uniqId, uniqByFirst, rankOfUniq;
        firstInst.alloc(nU); firstSorted.alloc(nU); uniqId.alloc(nU); uniqByFirst.alloc(nU); rankOfUniq.alloc(nU);
        hipLaunchKernelGGL(k_topo_edge_first, dim3(grid_of(n)), dim3(256), 0, s, n, kB.p, vB.p, headP1.p, firstInst.p);
        hipLaunchKernelGGL(k_topo_iota, dim3(grid_of(nU)), dim3(256), 0, s, (int64_t)nU, uniqId.p);
        RP(rocprim::radix_sort_pairs(nullptr, b, firstInst.p, firstSorted.p, uniqId.p, uniqByFirst.p, (size_t)nU, 0u, bits_for((uint64_t)n), s));
        ensureTmp(b);
        RP(rocprim::radix_sort_pairs(tmp.p, b, firstInst.p, firstSorted.p, uniqId.p, uniqByFirst.p, (size_t)nU, 0u, bits_for((uint64_t)n), s));
        hipLaunchKernelGGL(k_topo_rank, dim3(grid_of(nU)), dim3(256), 0, s, (int64_t)nU, uniqByFirst.p, rankOfUniq.p);
        dInst.alloc(n);
        hipLaunchKernelGGL(k_topo_edge_assign, dim3(grid_of(n)), dim3(256), 0, s, n, vB.p, headP1.p, rankOfUniq.p, dInst.p);
        RP(hipGetLastError());
    }
    if (dElemNodesOut && dNodePosOut && hostVertPos) {
        // the device copies of the node table and of the node positions: only the vertices cross the bus
        const int64_t nNode = nVert + nEdgeNodes;
        dElemNodesOut->alloc((size_t)nElem * (nv + (deg == 2 ? nedge : 0)));
        dNodePosOut->alloc((size_t)nNode * dim);
        MFH_HIP(hipMemcpyAsync(dNodePosOut->p, hostVertPos, (size_t)nVert * dim * sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_topo_node_tables, dim3(grid_of(nElem)), dim3(256), 0, s, nElem, dim, deg, nVert, dEv.p, dInst.p, dElemNodesOut->p, dNodePosOut->p);
        RP(hipGetLastError());
    }
    {   // boundary half-faces / half-edges
        const int64_t n = nElem * (dim == 3 ? 4 : 3);
        DBuf<uint64_t> kA, kB;
        DBuf<uint32_t> vA, vB, flag, posP1, out;
        DBuf<int> nm;
        kA.alloc(n); kB.alloc(n); vA.alloc(n); vB.alloc(n); flag.alloc(n); posP1.alloc(n); nm.alloc(1);
        nm.zero(s);
        size_t b = 0;
        if (wide) {
            // order by (lo, mid, hi): stable sort by hi, then stable sort by lo:mid
            DBuf<uint32_t> h32A, h32B;
            h32A.alloc(n); h32B.alloc(n);
            hipLaunchKernelGGL(k_topo_face_keys_wide, dim3(grid_of(n)), dim3(256), 0, s, n, dEv.p, h32A.p, vA.p);
            RP(rocprim::radix_sort_pairs(nullptr, b, h32A.p, h32B.p, vA.p, vB.p, (size_t)n, 0u, bits_for((uint64_t)nVert), s));
            ensureTmp(b);
            RP(rocprim::radix_sort_pairs(tmp.p, b, h32A.p, h32B.p, vA.p, vB.p, (size_t)n, 0u, bits_for((uint64_t)nVert), s));
            hipLaunchKernelGGL(k_topo_face_keys_lomid, dim3(grid_of(n)), dim3(256), 0, s, n, dEv.p, vB.p, kA.p);
            const unsigned eb = 32 + bits_for((uint64_t)nVert);
            RP(rocprim::radix_sort_pairs(nullptr, b, kA.p, kB.p, vB.p, vA.p, (size_t)n, 0u, eb, s));
            ensureTmp(b);
            RP(rocprim::radix_sort_pairs(tmp.p, b, kA.p, kB.p, vB.p, vA.p, (size_t)n, 0u, eb, s));
            MFH_HIP(hipMemcpyAsync(vB.p, vA.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));   // vB = sorted instances, as below
            hipLaunchKernelGGL(k_topo_face_flags_wide, dim3(grid_of(n)), dim3(256), 0, s, n, dEv.p, vB.p, flag.p, nm.p);
        } else {
        hipLaunchKernelGGL(k_topo_face_keys, dim3(grid_of(n)), dim3(256), 0, s, n, dim, dEv.p, kA.p, vA.p);
        const unsigned endBit = dim == 3 ? 63u : 32 + bits_for((uint64_t)nVert);
        RP(rocprim::radix_sort_pairs(nullptr, b, kA.p, kB.p, vA.p, vB.p, (size_t)n, 0u, endBit, s));
        ensureTmp(b);
        RP(rocprim::radix_sort_pairs(tmp.p, b, kA.p, kB.p, vA.p, vB.p, (size_t)n, 0u, endBit, s));
        hipLaunchKernelGGL(k_topo_face_flags, dim3(grid_of(n)), dim3(256), 0, s, n, kB.p, flag.p, nm.p);
        }
        RP(rocprim::inclusive_scan(nullptr, b, flag.p, posP1.p, (size_t)n, rocprim::plus<uint32_t>(), s));
        ensureTmp(b);
        RP(rocprim::inclusive_scan(tmp.p, b, flag.p, posP1.p, (size_t)n, rocprim::plus<uint32_t>(), s));
        uint32_t nB = 0;
        int nonManifold = 0;
        // everything up to here is queued: the caller's second batch of host work (stage 1: the node positions, whose size is known now)
        // runs while the device sorts the half-faces; the edge-node ranks come down behind it
        if (hostOverlap) hostOverlap(1, nEdgeNodes);
        if (deg == 2) dInst.download(instEdge.data(), instEdge.size(), s);
        MFH_HIP(hipMemcpyAsync(&nB, posP1.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        MFH_HIP(hipMemcpyAsync(&nonManifold, nm.p, sizeof(int), hipMemcpyDeviceToHost, s));
        MFH_HIP(hipStreamSynchronize(s));
        if (nonManifold) throw Error(MFH_ERR_INVALID, dim == 3 ? "Non-manifold input detected." : "Non-manifold edge detected");
        out.alloc(std::max<uint32_t>(nB, 1));
        hipLaunchKernelGGL(k_topo_face_compact, dim3(grid_of(n)), dim3(256), 0, s, n, vB.p, flag.p, posP1.p, out.p);
        RP(hipGetLastError());
        bdryInst.resize(nB);
        out.download(bdryInst.data(), nB, s);
    }
    return true;
}


// A locality-preserving order of the elements for the cluster operator: elements sorted by the Morton code of the CELL their
// centroid falls in (stable radix sort), so that a block of consecutive elements is a compact clump whatever order the caller's
// mesh has -- fewer distinct rows per block (LDS, x staging) and fewer interface partials than e.g. the 1 x 1 x 10.7-hex
// pencils that 256 consecutive elements of the reference's grid generator form. perm[new] = old; elemNodesOut is the
// connectivity in the new order.
void build_element_order_device(const HostMesh &m, const int32_t *dElemNodes, const double *dNodePos, hipStream_t s, DBuf<int32_t> &perm,
                                DBuf<int32_t> &elemNodesOut, int maxBlock, std::vector<int32_t> &blockStart) {
    const bool timing = getenv("MFH_MFC_TIMING") != nullptr;
    double tp = now_ms();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        const double t = now_ms();
        fprintf(stderr, "[element order]  %-40s %8.2f ms\n", what, t - tp);
        tp = t;
    };
    const int dim = m.dim;
    // bounding box of the nodes: a reduction on the device copy (minima / maxima are exact in any order; the host loop over the 57.6 M
    // nodes of a 119^3 grid took 130 ms)
    double lo[3], hi[3];
    bounding_box_device(dim, m.nNode, dNodePos, s, lo, hi);
    lap("bounding box");
    // Cells of ~24 elements (one hex of the reference's grid generator): the elements of a cell share a key and keep their
    // original order (stable sort), the cells follow the Z-curve. Finer keys were measured WORSE than the generator's own
    // order (interface partials 8.64 M vs 7.45 M at config 3): the curve then cuts through the hexes and the blocks get ragged.
    const double top = dim == 3 ? 2097151.0 : 2147483647.0;
    double vol = 1.0;
    for (int a = 0; a < dim; ++a) vol *= std::max(hi[a] - lo[a], 1e-300);
    const double h = std::pow(vol * (dim == 3 ? 24.0 : 4.0) / (double)std::max<int64_t>(m.nElem, 1), 1.0 / dim);
    double sc[3] = {0, 0, 0};
    for (int a = 0; a < dim; ++a) sc[a] = hi[a] > lo[a] ? std::min(1.0 / h, top / (hi[a] - lo[a])) : 0.0;
    DBuf<uint64_t> keyA, keyB;
    DBuf<uint32_t> valA, valB;
    keyA.alloc((size_t)m.nElem); keyB.alloc((size_t)m.nElem); valA.alloc((size_t)m.nElem); valB.alloc((size_t)m.nElem);
    hipLaunchKernelGGL(k_elem_morton, dim3(grid_of(m.nElem)), dim3(256), 0, s, m.nElem, m.npe, dim, dElemNodes, dNodePos, lo[0], lo[1], lo[2], sc[0], sc[1],
                       sc[2], keyA.p, valA.p);
    RP(hipGetLastError());
    size_t tmpBytes = 0;
    RP(rocprim::radix_sort_pairs(nullptr, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)m.nElem, 0u, 63u, s));
    DBuf<char> tmp;
    tmp.alloc(tmpBytes + 16);
    RP(rocprim::radix_sort_pairs(tmp.p, tmpBytes, keyA.p, keyB.p, valA.p, valB.p, (size_t)m.nElem, 0u, 63u, s));
    lap("keys + sort");
    // blocks = whole cells packed greedily along the curve, at most maxBlock elements each (a block boundary inside a cell
    // would put that cell's shared rows on the interface); a cell larger than a block is split
    // The greedy packing is sequential over the CELLS (1.7 M at 119^3), not over the elements: the device compacts the first element of
    // every cell, the host walks those (the sorted keys themselves, 323 MB at 119^3, stay on the device: download + scan took 95 ms)
    {
        DBuf<uint32_t> cellP1;
        DBuf<int32_t> cellStart;
        cellP1.alloc((size_t)m.nElem);
        hipLaunchKernelGGL(k_sym_heads, dim3(grid_of(m.nElem)), dim3(256), 0, s, m.nElem, keyB.p, cellP1.p, 0u);
        size_t scanBytes = 0;
        RP(rocprim::inclusive_scan(nullptr, scanBytes, cellP1.p, cellP1.p, (size_t)m.nElem, rocprim::plus<uint32_t>(), s));
        if (scanBytes + 16 > tmp.n) tmp.alloc(scanBytes + 16);
        RP(rocprim::inclusive_scan(tmp.p, scanBytes, cellP1.p, cellP1.p, (size_t)m.nElem, rocprim::plus<uint32_t>(), s));
        uint32_t nCells = 0;
        MFH_HIP(hipMemcpyAsync(&nCells, cellP1.p + (m.nElem - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        MFH_HIP(hipStreamSynchronize(s));
        cellStart.alloc((size_t)nCells);
        hipLaunchKernelGGL(k_cell_starts, dim3(grid_of(m.nElem)), dim3(256), 0, s, m.nElem, keyB.p, cellP1.p, cellStart.p);
        RP(hipGetLastError());
        std::vector<int32_t> cs((size_t)nCells + 1);
        cellStart.download(cs.data(), (size_t)nCells, s);
        cs[(size_t)nCells] = (int32_t)m.nElem;
        blockStart.assign(1, 0);
        int64_t cur = 0;                                   // elements in the open block
        for (size_t ci = 0; ci < (size_t)nCells; ++ci) {
            int64_t e = cs[ci], len = cs[ci + 1] - cs[ci];
            if (cur > 0 && cur + len > maxBlock) { blockStart.push_back((int32_t)e); cur = 0; }
            while (len > maxBlock) { e += maxBlock; len -= maxBlock; blockStart.push_back((int32_t)e); }
            cur += len;
        }
        blockStart.push_back((int32_t)m.nElem);
    }
    lap("cell starts + block scan");
    perm.alloc((size_t)m.nElem);
    elemNodesOut.alloc((size_t)m.nElem * m.npe);
    hipLaunchKernelGGL(k_permute_rows_i32, dim3(grid_of(m.nElem * m.npe)), dim3(256), 0, s, m.nElem, m.npe, valB.p, dElemNodes, elemNodesOut.p, perm.p);
    RP(hipGetLastError());
    lap("permuted connectivity");
    MFH_HIP(hipStreamSynchronize(s));
}

} // namespace mfh
