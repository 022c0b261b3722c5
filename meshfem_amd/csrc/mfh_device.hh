// Device-side helpers shared by the kernel translation units (mfh_kernels.hip: element embedding, assembly, operators and
// element-level post-processing; mfh_kernels_solver.hip: diagonal blocks, two-level preconditioner, dense coarse inverse and
// the PCG vector kernels).
#pragma once
#include "mfh_internal.hh"
#include <algorithm>

namespace mfh { namespace k {

#define DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
// (tile geometry: MFH_TILE_LOG / MFH_TILE_PAD, mfh_internal.hh)
DEV int64_t tiled_index(int64_t slot, int c, int NB) {
    return (slot >> MFH_TILE_LOG) * (((int64_t)NB << MFH_TILE_LOG) + MFH_TILE_PAD) + ((int64_t)c << MFH_TILE_LOG) + (slot & ((1 << MFH_TILE_LOG) - 1));
}

DEV double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-wide sum of up to 3 values; result valid in thread 0. blockDim.x == 256.
template <int NV>
DEV void block_sum(double (&v)[NV], double *lds /* >= 4*NV doubles */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) lds[w * NV + k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = lds[k] + lds[NV + k] + lds[2 * NV + k] + lds[3 * NV + k];
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Global sums of workgroup partials (dot products of the PCG). Default: one unsafeAtomicAdd per workgroup onto the target -- the
// order in which the workgroups arrive, and with it the last bits of the sum, changes from run to run. Option "deterministic"
// (d.partials != null): every workgroup stores its partials with plain stores, workgroup 0 also the addresses of the targets, and the
// kernel ends; a one-workgroup kernel launched right behind it (k_det_finish, launch_det_finish) adds the partials up in workgroup
// order with a fixed tree and adds the totals onto the targets -- the same bits on every run. No fence inside the producing kernel: a
// device-scope release per workgroup writes back and invalidates the whole L2 of its XCD (measured: k_pcg_update 0.22 -> 0.39 ms,
// k_mf_cluster 0.51 -> 1.49 ms with a last-workgroup-done scheme); the kernel boundary does it once. Layout of the scratch:
// [0] number of sums (0 = the producer was gated off: nothing to do), [1] workgroups, [2..5] target addresses, [8 + b 4 + k] partials.
// v: valid in thread 0. target[k] == nullptr: that sum is not wanted.
// ------------------------------------------------------------------------------------------------
constexpr int DET_HEADER = 8;
template <int NV>
DEV void commit_sums(double (&v)[NV], double *const (&target)[NV], const DetBuf &d, double * /* lds: unused */) {
    if (threadIdx.x != 0) return;
    if (!d.partials) {
#pragma unroll
        for (int k = 0; k < NV; ++k) if (target[k]) unsafeAtomicAdd(target[k], v[k]);
        return;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) d.partials[DET_HEADER + (int64_t)blockIdx.x * 4 + k] = v[k];
    if (blockIdx.x == 0) {
        unsigned long long *h = reinterpret_cast<unsigned long long *>(d.partials);
        h[0] = NV; h[1] = gridDim.x;
#pragma unroll
        for (int k = 0; k < NV; ++k) h[2 + k] = (unsigned long long)target[k];
    }
}

// XCD-aware work mapping. Workgroups are dispatched round-robin over the 8 XCDs (workgroup b runs on XCD b % 8) and
// every XCD has its own L2, so neighbouring work items (row chunks, element groups: they share element records, gather
// lists and x entries) should run on the SAME XCD: XCD x gets the x-th contiguous eighth of the n items.
// One workgroup per item: a bijection of [0, n).
DEV int64_t xcd_item(int64_t b, int64_t n) {
    const int64_t q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
    return x * q + (x < r ? x : r) + k;
}
// Grouped variant: XCD x takes runs of G consecutive items, all eight XCDs staying inside one window of 8 G items -- neighbouring
// items (which share element records / x entries) meet in one L2 while the HBM stream stays sequential across the chip.
// A bijection of [0, n): the last, partial window keeps the identity.
DEV int64_t xcd_group_item(int64_t b, int64_t n, int G) {
    const int64_t win = (int64_t)8 * G, w = b / win;
    if ((w + 1) * win > n) return b;
    const int64_t r = b - w * win;
    return w * win + (r & 7) * G + (r >> 3);
}
// Persistent workgroups (gridDim.x a multiple of 8): the items of XCD x are [begin, end), visited with stride gridDim.x / 8
// starting at begin + blockIdx.x / 8.
DEV void xcd_span(int64_t n, int64_t &first, int64_t &end, int64_t &stride) {
    const int64_t q = n >> 3, r = n & 7, x = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int64_t begin = x * q + (x < r ? x : r);
    end = begin + q + (x < r ? 1 : 0);
    first = begin + k;
    stride = gridDim.x >> 3;
}

// Flattened symmetric index (Flattening.hh:47-60): 3D xx,yy,zz,yz,xz,xy ; 2D xx,yy,xy
template <int DIM>
DEV constexpr int flat_idx(int i, int j) { return i == j ? i : (DIM * (DIM + 1) / 2 - i - j); }
// index into the packed upper triangle (row-major) of the flatLen x flatLen matrix D
template <int DIM>
DEV constexpr int dpack(int r, int c) {
    constexpr int n = DIM * (DIM + 1) / 2;
    int a = r <= c ? r : c, b = r <= c ? c : r;
    return a * n - a * (a - 1) / 2 + (b - a);
}

// orthotropic record (MAT_ORTHO): g[13 ..] = packed upper triangle of the DIM x DIM normal block of D, then the DIM(DIM-1)/2
// shear stiffnesses in flattened order (yz, xz, xy | xy)
template <int DIM> DEV constexpr int npack(int a, int b) {
    const int lo = a <= b ? a : b, hi = a <= b ? b : a;
    return lo * DIM - lo * (lo - 1) / 2 + (hi - lo);
}
template <int DIM> DEV constexpr int ortho_shear_offset() { return 13 + DIM * (DIM + 1) / 2; }

// the six distinct quadrature pair coefficients (named scalars: an indexed array would go to scratch)
struct PairConst { double vv_eq, vv_ne, ve_eq, ve_ne, ee_eq, ee_ne; };

// support vertices of node i: grad phi_i = alpha gl[s] + beta gl[t]   (EmbeddedElement.hh:315-332)
// packed 4-bit tables: vertex nodes s=t=i; edge node k: s=edgeStart[k], t=edgeEnd[k] (Simplex.hh:43-44)
template <int DIM, int DEG> DEV int sup_s(int i) {
    if (DEG == 1) return i;
    if (DIM == 3) return (int)((0x1202103210ull >> (4 * i)) & 0xf);   // nodes 0..9: 0,1,2,3,0,1,2,0,2,1
    return (int)((0x210210ull >> (4 * i)) & 0xf);                     // nodes 0..5: 0,1,2,0,1,2
}
template <int DIM, int DEG> DEV int sup_t(int i) {
    if (DEG == 1) return i;
    if (DIM == 3) return (int)((0x3330213210ull >> (4 * i)) & 0xf);   // 0,1,2,3,1,2,0,3,3,3
    return (int)((0x021210ull >> (4 * i)) & 0xf);                     // 0,1,2,1,2,0
}


// ---- launch helpers
static inline int grid_for(int64_t n, int cap = 2048) {
    int64_t g = (n + 255) / 256;
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}
#define CHECK_LAUNCH() MFH_HIP(hipGetLastError())

}} // namespace mfh::k
