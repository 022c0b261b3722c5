// HIP kernels for gfx950 (MI355X, CDNA4), solver side: inverse diagonal blocks, block-Jacobi / Jacobi application, the
// two-level preconditioner (restriction, Galerkin product, prolongation), the dense SPD inverse of its coarse operator and
// the fused PCG vector kernels. All arithmetic is FP64. See DESIGN.md sections 4.4b and 4.5.
#include "mfh_device.hh"

namespace mfh { namespace k {

// ------------------------------------------------------------------------------------------------
// small dense helpers
// ------------------------------------------------------------------------------------------------
template <int DIM> DEV void invert_block(const double *A, double *Inv) {
    if (DIM == 1) {
        Inv[0] = 1.0 / A[0];
    } else if (DIM == 2) {
        const double det = A[0] * A[3] - A[1] * A[2];
        Inv[0] = A[3] / det; Inv[1] = -A[1] / det; Inv[2] = -A[2] / det; Inv[3] = A[0] / det;
    } else {
        const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
        const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
        Inv[0] = c00 / det; Inv[1] = (A[2] * A[7] - A[1] * A[8]) / det; Inv[2] = (A[1] * A[5] - A[2] * A[4]) / det;
        Inv[3] = c01 / det; Inv[4] = (A[0] * A[8] - A[2] * A[6]) / det; Inv[5] = (A[2] * A[3] - A[0] * A[5]) / det;
        Inv[6] = c02 / det; Inv[7] = (A[1] * A[6] - A[0] * A[7]) / det; Inv[8] = (A[0] * A[4] - A[1] * A[3]) / det;
    }
}

// Inverse of the diagonal blocks of the constrained operator P K P + (I-P): rows/cols of fixed
// components are replaced by identity before inversion. kind: 0 block-Jacobi, 1 Jacobi, 2 identity.
template <int DIM>
__global__ void __launch_bounds__(256) k_diag_inv(int64_t nRows, const int32_t *__restrict__ rowPtr,
                                                  const int32_t *__restrict__ colIdx, const double *__restrict__ vals,
                                                  const uint8_t *__restrict__ fixedMask, int kind, double *__restrict__ dinv) {
    constexpr int NB = DIM * DIM;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nRows) return;
    double A[NB], Inv[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) A[c] = (c % (DIM + 1) == 0) ? 1.0 : 0.0;
    int lo = rowPtr[r], hi = rowPtr[r + 1];
    while (lo < hi) {   // columns are sorted within a row
        const int mid = lo + ((hi - lo) >> 1);   // (lo + hi) overflows int beyond 2^30 blocks (found at 31.9 M P2 tets: 1.24e9 blocks)
        const int cv = colIdx[mid];
        if (cv == r) {
#pragma unroll
            for (int c = 0; c < NB; ++c) A[c] = vals[tiled_index(mid, c, NB)];
            break;
        }
        if (cv < r) lo = mid + 1; else hi = mid;
    }
    if (fixedMask) {
#pragma unroll
        for (int c = 0; c < DIM; ++c)
            if (fixedMask[r * DIM + c]) {
#pragma unroll
                for (int d = 0; d < DIM; ++d) { A[c * DIM + d] = 0.0; A[d * DIM + c] = 0.0; }
                A[c * DIM + c] = 1.0;
            }
    }
    if (kind == 0) invert_block<DIM>(A, Inv);
    else {
#pragma unroll
        for (int c = 0; c < NB; ++c) Inv[c] = 0.0;
#pragma unroll
        for (int c = 0; c < DIM; ++c) Inv[c * DIM + c] = kind == 1 ? 1.0 / A[c * DIM + c] : 1.0;
    }
    // the inverse of a symmetric block is symmetric: stored packed (flat symmetric index), 6 instead of 9 values in 3D
    constexpr int NS = DIM * (DIM + 1) / 2;
#pragma unroll
    for (int c = 0; c < DIM; ++c)
#pragma unroll
        for (int d = c; d < DIM; ++d) dinv[r * NS + flat_idx<DIM>(c, d)] = 0.5 * (Inv[c * DIM + d] + Inv[d * DIM + c]);
}

// z = D^-1 r with the symmetric-packed inverse diagonal block (DIM (DIM+1)/2 values per block row)
template <int DIM> DEV void apply_block(const double *__restrict__ Dm, const double *r, double *z) {
    constexpr int NS = DIM * (DIM + 1) / 2;
    double m[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) m[q] = Dm[q];
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
        double v = 0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) v += m[flat_idx<DIM>(c, d)] * r[d];
        z[c] = v;
    }
}

// the packed inverse diagonal blocks of TWO consecutive rows (2 NS values, contiguous) in the widest aligned accesses; the FP32 copy (the multigrid
// smoother's fused kernels, option mg_dinv_fp32) is widened on the way in
template <int NS> DEV void load_dinv_pair(const double *__restrict__ p, double (&dm)[2 * NS]) {
    const double2 *d2 = reinterpret_cast<const double2 *>(p);
#pragma unroll
    for (int c = 0; c < NS; ++c) { const double2 t = d2[c]; dm[2 * c] = t.x; dm[2 * c + 1] = t.y; }
}
template <int NS> DEV void load_dinv_pair(const float *__restrict__ p, double (&dm)[2 * NS]) {
    if (NS % 2 == 0) {          // 2 NS floats = NS / 2 sixteen-byte pieces (3D: 48 bytes per row pair)
        const float4 *d4 = reinterpret_cast<const float4 *>(p);
#pragma unroll
        for (int c = 0; c < NS / 2; ++c) { const float4 t = d4[c]; dm[4 * c] = t.x; dm[4 * c + 1] = t.y; dm[4 * c + 2] = t.z; dm[4 * c + 3] = t.w; }
    } else {
        const float2 *d2 = reinterpret_cast<const float2 *>(p);
#pragma unroll
        for (int c = 0; c < NS; ++c) { const float2 t = d2[c]; dm[2 * c] = t.x; dm[2 * c + 1] = t.y; }
    }
}
template <int DIM, class DinvT> DEV void apply_block_t(const DinvT *__restrict__ Dm, const double *r, double *z) {
    constexpr int NS = DIM * (DIM + 1) / 2;
    double m[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) m[q] = (double)Dm[q];
    apply_block<DIM>(m, r, z);
}

template <int DIM>
__global__ void __launch_bounds__(256) k_precond(int64_t nRows, const double *__restrict__ dinv, const double *__restrict__ r,
                                                 double *__restrict__ z) {
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nRows; n += (int64_t)gridDim.x * 256) {
        double rv[DIM], zv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) rv[c] = r[n * DIM + c];
        apply_block<DIM>(dinv + n * (DIM * (DIM + 1) / 2), rv, zv);
#pragma unroll
        for (int c = 0; c < DIM; ++c) z[n * DIM + c] = zv[c];
    }
}

// ------------------------------------------------------------------------------------------------
// Two-level preconditioner  M^-1 = D^-1 + Z (Z^T K Z)^-1 Z^T  (mfh_twolevel.cpp): Z = rigid-body
// modes of every aggregate (translations, and rotations about the aggregate centroid scaled by 1/H),
// zero on fixed variables. Z is never stored: a mode's value at a DoF follows from its relative position.
// ------------------------------------------------------------------------------------------------
template <int DIM> DEV double tl_mode(int k, int c, const double *rp) {
    const double rx = rp[0], ry = rp[1], rz = rp[2];
    if (k < DIM) return k == c ? 1.0 : 0.0;
    if (DIM == 2) return c == 0 ? -ry : rx;
    // rotation about axis k-3: u = e_axis x r
    if (k == 3) return c == 1 ? -rz : (c == 2 ? ry : 0.0);
    if (k == 4) return c == 0 ? rz : (c == 2 ? -rx : 0.0);
    return c == 0 ? -ry : (c == 1 ? rx : 0.0);
}

// probing vector: sum over the aggregates of one colour of their mode `mode`
template <int DIM>
__global__ void __launch_bounds__(256) k_tl_fill(TLArgs t, const int32_t *__restrict__ colorOfAgg, int color, int mode, double *__restrict__ v) {
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < t.nDoF; n += (int64_t)gridDim.x * 256) {
        const int a = t.aggOfDof[n];
        const bool on = colorOfAgg[a] == color;
        double rp[3] = {t.relPos[n * 3], t.relPos[n * 3 + 1], t.relPos[n * 3 + 2]};
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const bool fixed = t.fixedMask && t.fixedMask[n * DIM + c];
            v[n * DIM + c] = (on && !fixed) ? tl_mode<DIM>(mode, c, rp) : 0.0;
        }
    }
}

// rc[a*nModes + k] = sum_{DoFs n of aggregate a} z_{a,k}(n) . w(n); one workgroup per aggregate (the ~1 000 aggregates of ~1 000 nodes of the
// two-level preconditioner)
template <int DIM>
__global__ void __launch_bounds__(256) k_tl_restrict(TLArgs t, const int32_t *__restrict__ aggPtr, const int32_t *__restrict__ dofsByAgg,
                                                     const double *__restrict__ w, double *__restrict__ rc) {
    __shared__ double red[4 * 6];
    const int a = blockIdx.x;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int q = aggPtr[a] + threadIdx.x; q < aggPtr[a + 1]; q += 256) {
        const int64_t n = dofsByAgg[q];
        double rp[3] = {t.relPos[n * 3], t.relPos[n * 3 + 1], t.relPos[n * 3 + 2]};
        double wv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) wv[c] = (t.fixedMask && t.fixedMask[n * DIM + c]) ? 0.0 : w[n * DIM + c];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (k >= t.nModes) break;
            double s = 0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) s += tl_mode<DIM>(k, c, rp) * wv[c];
            acc[k] += s;
        }
    }
    const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 6; ++k) red[wv_ * 6 + k] = acc[k];
    __syncthreads();
    if (threadIdx.x < t.nModes) rc[a * t.nModes + threadIdx.x] = red[threadIdx.x] + red[6 + threadIdx.x] + red[12 + threadIdx.x] + red[18 + threadIdx.x];
}

// The same restriction with one WAVE per aggregate, four aggregates per workgroup: the first aggregate level of the multigrid hierarchy holds
// ~32 vertices per aggregate (32 768 aggregates at config 3), where a 256-lane workgroup per aggregate keeps an eighth of its lanes busy and pays a
// barrier for a sum one wave holds (113 -> us at config 3, rocprofv3).
template <int DIM>
__global__ void __launch_bounds__(256) k_tl_restrict_wave(TLArgs t, const int32_t *__restrict__ aggPtr, const int32_t *__restrict__ dofsByAgg,
                                                          const double *__restrict__ w, double *__restrict__ rc) {
    const int lane = threadIdx.x & 63;
    const int64_t a = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= t.nAgg) return;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int q = aggPtr[a] + lane; q < aggPtr[a + 1]; q += 64) {
        const int64_t n = dofsByAgg[q];
        double rp[3] = {t.relPos[n * 3], t.relPos[n * 3 + 1], t.relPos[n * 3 + 2]};
        double wv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) wv[c] = (t.fixedMask && t.fixedMask[n * DIM + c]) ? 0.0 : w[n * DIM + c];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (k >= t.nModes) break;
            double s = 0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) s += tl_mode<DIM>(k, c, rp) * wv[c];
            acc[k] += s;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < t.nModes) rc[a * t.nModes + k] = acc[k];
}

// the wave-per-aggregate restriction for NR interleaved vectors (w entry ((n NR + kr) DIM + c), result [coarse index][NR]): a DoF's relative
// position and mask are fetched once for all NR
template <int DIM, int NR>
__global__ void __launch_bounds__(256) k_tl_restrict_wave_nr(TLArgs t, const int32_t *__restrict__ aggPtr, const int32_t *__restrict__ dofsByAgg,
                                                             const double *__restrict__ w, double *__restrict__ rc) {
    const int lane = threadIdx.x & 63;
    const int64_t a = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (a >= t.nAgg) return;
    double acc[NR][6];
#pragma unroll
    for (int kr = 0; kr < NR; ++kr)
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[kr][k] = 0.0;
    for (int q = aggPtr[a] + lane; q < aggPtr[a + 1]; q += 64) {
        const int64_t n = dofsByAgg[q];
        double rp[3] = {t.relPos[n * 3], t.relPos[n * 3 + 1], t.relPos[n * 3 + 2]};
        bool fx[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) fx[c] = t.fixedMask && t.fixedMask[n * DIM + c];
#pragma unroll
        for (int kr = 0; kr < NR; ++kr) {
            double wv[DIM];
#pragma unroll
            for (int c = 0; c < DIM; ++c) wv[c] = fx[c] ? 0.0 : w[(n * NR + kr) * DIM + c];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (k >= t.nModes) break;
                double s = 0;
#pragma unroll
                for (int c = 0; c < DIM; ++c) s += tl_mode<DIM>(k, c, rp) * wv[c];
                acc[kr][k] += s;
            }
        }
    }
#pragma unroll
    for (int kr = 0; kr < NR; ++kr)
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[kr][k] = wave_sum(acc[kr][k]);
    if (lane == 0)
#pragma unroll
        for (int kr = 0; kr < NR; ++kr)
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (k < t.nModes) rc[(a * t.nModes + k) * NR + kr] = acc[kr][k];
}

// coarse operator entries from one probe: Ac[(b,l), (nbr(b,colour), mode)] = R[(b,l)]
__global__ void __launch_bounds__(256) k_tl_scatter(int nAgg, int nModes, int nColor, const int32_t *__restrict__ nbrOfColor, int color,
                                                    int mode, const double *__restrict__ R, double *__restrict__ Ac) {
    const int64_t m = (int64_t)nAgg * nModes;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < m; k += (int64_t)gridDim.x * 256) {
        const int b = (int)(k / nModes);
        const int a = nbrOfColor[(int64_t)b * nColor + color];
        if (a >= 0) Ac[k * m + (int64_t)a * nModes + mode] = R[k];
    }
}

// Coarse operator in ONE pass over the assembled K (Galerkin product Z^T K Z): one wave per block
// row; a lane takes a block K_rc, forms T[k][l] = z_k(r)^T K_rc z_l(c) for the modes of the two
// aggregates and adds it to Ac[(agg r, k), (agg c, l)]. Blocks inside one aggregate (the vast
// majority) are summed across the wave first, so only one set of atomics per row reaches memory.
// Replaces 3^dim * nModes probing SpMVs (162 in 3D).
template <int DIM>
__global__ void __launch_bounds__(256) k_tl_rap(TLArgs t, int64_t nRows, const int32_t *__restrict__ rowPtr, const int32_t *__restrict__ colIdx,
                                                const double *__restrict__ vals, double *__restrict__ Ac) {
    constexpr int NB = DIM * DIM;
    constexpr int NM = DIM == 3 ? 6 : 3;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nWaves = ((int64_t)gridDim.x * 256) >> 6;
    const int64_t m = (int64_t)t.nAgg * NM;
    for (int64_t r = wave; r < nRows; r += nWaves) {
        const int a = t.aggOfDof[r];
        const double rpr[3] = {t.relPos[r * 3], t.relPos[r * 3 + 1], t.relPos[r * 3 + 2]};
        double zr[NM][DIM];
#pragma unroll
        for (int k = 0; k < NM; ++k)
#pragma unroll
            for (int x = 0; x < DIM; ++x) zr[k][x] = (t.fixedMask && t.fixedMask[r * DIM + x]) ? 0.0 : tl_mode<DIM>(k, x, rpr);
        double acc[NM * NM];
#pragma unroll
        for (int q = 0; q < NM * NM; ++q) acc[q] = 0.0;
        for (int s = rowPtr[r] + lane; s < rowPtr[r + 1]; s += 64) {
            const int64_t c = colIdx[s];
            const int b = t.aggOfDof[c];
            const double rpc[3] = {t.relPos[c * 3], t.relPos[c * 3 + 1], t.relPos[c * 3 + 2]};
            double K[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) K[q] = vals[tiled_index(s, q, NB)];
            double T[NM * NM];
#pragma unroll
            for (int l = 0; l < NM; ++l) {
                double zc[DIM], w[DIM];
#pragma unroll
                for (int y = 0; y < DIM; ++y) zc[y] = (t.fixedMask && t.fixedMask[c * DIM + y]) ? 0.0 : tl_mode<DIM>(l, y, rpc);
#pragma unroll
                for (int x = 0; x < DIM; ++x) {
                    double v = 0;
#pragma unroll
                    for (int y = 0; y < DIM; ++y) v += K[x * DIM + y] * zc[y];
                    w[x] = v;
                }
#pragma unroll
                for (int k = 0; k < NM; ++k) {
                    double v = 0;
#pragma unroll
                    for (int x = 0; x < DIM; ++x) v += zr[k][x] * w[x];
                    T[k * NM + l] = v;
                }
            }
            if (b == a) {
#pragma unroll
                for (int q = 0; q < NM * NM; ++q) acc[q] += T[q];
            } else {
#pragma unroll
                for (int k = 0; k < NM; ++k)
#pragma unroll
                    for (int l = 0; l < NM; ++l) unsafeAtomicAdd(&Ac[((int64_t)a * NM + k) * m + (int64_t)b * NM + l], T[k * NM + l]);
            }
        }
#pragma unroll
        for (int q = 0; q < NM * NM; ++q) acc[q] = wave_sum(acc[q]);
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < NM; ++k)
#pragma unroll
                for (int l = 0; l < NM; ++l) unsafeAtomicAdd(&Ac[((int64_t)a * NM + k) * m + (int64_t)a * NM + l], acc[k * NM + l]);
    }
}

// offset between two bins of a lattice that is periodic with nb bins along this axis (nb <= 2: every bin is adjacent anyway)
DEV int lattice_wrap(int d, int nb) { return nb > 2 ? (d == nb - 1 ? -1 : (d == 1 - nb ? 1 : d)) : d; }

// Galerkin product, aggregate-centric: one workgroup (8 waves) per ROW aggregate walks that aggregate's rows; the rows
// (a, .) of the coarse operator are written by this workgroup alone, so nothing needs a global atomic:
//   * blocks whose column lies in the same aggregate (the vast majority) are summed in registers over ALL rows of the wave;
//   * blocks reaching a lattice neighbour go to an LDS table [27][NM*NM] (ds_add_f64);
//   * anything else (no lattice information, non-adjacent aggregates) falls back to a global atomic.
// The per-row version above pays 36 atomics per row on the same few addresses (7 400 rows of an aggregate hammer one
// 6x6 block): 68 ms at config 3; this one streams K once.
template <int DIM>
__global__ void __launch_bounds__(512) k_tl_rap_agg(TLArgs t, const int32_t *__restrict__ aggPtr, const int32_t *__restrict__ dofsByAgg,
                                                     const int32_t *__restrict__ binCoord /* nAgg x 3, may be null */,
                                                     const int32_t *__restrict__ rowPtr, const int32_t *__restrict__ colIdx,
                                                     const double *__restrict__ vals, double *__restrict__ Ac, int upperOnly, int64_t nOwnedRows,
                                                     double *__restrict__ stencil, int *__restrict__ farCount, int wrapX, int wrapY, int wrapZ, int det) {
    // wrapA > 2: the lattice is periodic along that axis with wrapA bins (periodic DoF maps: elements at the seam couple the first and
    // the last bin), offsets +-(wrapA - 1) count as -+1
    // stencil != null (multigrid hierarchy, mfh_multigrid.cpp): the rows (a, .) go to the lattice-stencil storage
    // stencil[(a NSLOT + slot) NM^2 + k NM + l] (slot = the neighbour's lattice offset, centre = the aggregate itself) instead of the
    // dense matrix; a block reaching a non-adjacent aggregate cannot be stored there and is counted in farCount (the caller coarsens less).
    // upperOnly: only the blocks (r, c >= r) are stored. A stored off-diagonal block inside the aggregate also contributes its
    // transpose to the diagonal 6 x 6 block; between two aggregates this kernel writes the partial sums U[a][b] over the stored
    // blocks and k_tl_mirror_upper completes them (Ac[a][b] = U[a][b] + U[b][a]^T). On a row-partitioned context the blocks
    // (owned row, halo column) are stored by BOTH ranks that share them (halo columns are numbered after the owned rows): each
    // counts them half, and the all-reduce of the ranks' mirrored partial sums gives the whole.
    constexpr int NB = DIM * DIM;
    constexpr int NM = DIM == 3 ? 6 : 3;
    constexpr int NSLOT = DIM == 3 ? 27 : 9;
    // det (option "deterministic"): the waves of the workgroup add into neighbour tables of their OWN, summed in wave order at the end --
    // between waves the order of LDS atomics is a matter of timing (inside one wave the hardware applies the lanes in a fixed order)
    extern __shared__ __attribute__((aligned(16))) double rapLds[];   // neighbour tables [det ? nWaves : 1][NSLOT NM^2] + diagRed [16 NM^2]
    __shared__ int nbrAgg[NSLOT];
    const int a = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    const int nTab = det ? nWaves : 1;
    double *nbrAll = rapLds;
    double *nbr = rapLds + (det ? wave : 0) * (NSLOT * NM * NM);
    double *diagRed = rapLds + nTab * (NSLOT * NM * NM);
    const int64_t m = (int64_t)t.nAgg * NM;
    for (int q = threadIdx.x; q < nTab * NSLOT * NM * NM; q += blockDim.x) nbrAll[q] = 0.0;
    if (threadIdx.x < NSLOT) nbrAgg[threadIdx.x] = -1;
    __syncthreads();
    int ca[3] = {0, 0, 0};
    if (binCoord) { ca[0] = binCoord[a * 3]; ca[1] = binCoord[a * 3 + 1]; ca[2] = binCoord[a * 3 + 2]; }
    double acc[NM * NM];
#pragma unroll
    for (int q = 0; q < NM * NM; ++q) acc[q] = 0.0;
    for (int p = aggPtr[a] + wave; p < aggPtr[a + 1]; p += nWaves) {
        const int64_t r = dofsByAgg[p];
        const double rpr[3] = {t.relPos[r * 3], t.relPos[r * 3 + 1], t.relPos[r * 3 + 2]};
        double zr[NM][DIM];
#pragma unroll
        for (int k = 0; k < NM; ++k)
#pragma unroll
            for (int x = 0; x < DIM; ++x) zr[k][x] = (t.fixedMask && t.fixedMask[r * DIM + x]) ? 0.0 : tl_mode<DIM>(k, x, rpr);
        for (int s = rowPtr[r] + lane; s < rowPtr[r + 1]; s += 64) {
            const int64_t c = colIdx[s];
            const int b = t.aggOfDof[c];
            const double rpc[3] = {t.relPos[c * 3], t.relPos[c * 3 + 1], t.relPos[c * 3 + 2]};
            double K[NB];
            const double wgt = (upperOnly && c >= nOwnedRows) ? 0.5 : 1.0;
#pragma unroll
            for (int q = 0; q < NB; ++q) K[q] = wgt * vals[tiled_index(s, q, NB)];
            double T[NM * NM];
#pragma unroll
            for (int l = 0; l < NM; ++l) {
                double zc[DIM], w[DIM];
#pragma unroll
                for (int y = 0; y < DIM; ++y) zc[y] = (t.fixedMask && t.fixedMask[c * DIM + y]) ? 0.0 : tl_mode<DIM>(l, y, rpc);
#pragma unroll
                for (int x = 0; x < DIM; ++x) {
                    double v = 0;
#pragma unroll
                    for (int y = 0; y < DIM; ++y) v += K[x * DIM + y] * zc[y];
                    w[x] = v;
                }
#pragma unroll
                for (int k = 0; k < NM; ++k) {
                    double v = 0;
#pragma unroll
                    for (int x = 0; x < DIM; ++x) v += zr[k][x] * w[x];
                    T[k * NM + l] = v;
                }
            }
            if (b == a) {
#pragma unroll
                for (int q = 0; q < NM * NM; ++q) acc[q] += T[q];
                if (upperOnly && c != r) {
#pragma unroll
                    for (int q = 0; q < NM * NM; ++q) acc[q] += T[(q % NM) * NM + q / NM];
                }
                continue;
            }
            int slot = -1;
            if (binCoord) {
                int dx = binCoord[b * 3] - ca[0], dy = binCoord[b * 3 + 1] - ca[1], dz = binCoord[b * 3 + 2] - ca[2];
                dx = lattice_wrap(dx, wrapX); dy = lattice_wrap(dy, wrapY); dz = lattice_wrap(dz, wrapZ);
                if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1) slot = (dx + 1) + 3 * (dy + 1) + (DIM == 3 ? 9 * (dz + 1) : 0);
            }
            if (slot >= 0) {
                nbrAgg[slot] = b;     // every writer stores the same value
#pragma unroll
                for (int q = 0; q < NM * NM; ++q) unsafeAtomicAdd(&nbr[slot * NM * NM + q], T[q]);
            } else if (stencil) {
                atomicAdd(farCount, 1);
            } else {
#pragma unroll
                for (int k = 0; k < NM; ++k)
#pragma unroll
                    for (int l = 0; l < NM; ++l) unsafeAtomicAdd(&Ac[((int64_t)a * NM + k) * m + (int64_t)b * NM + l], T[k * NM + l]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NM * NM; ++q) acc[q] = wave_sum(acc[q]);
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < NM * NM; ++q) diagRed[wave * NM * NM + q] = acc[q];
    __syncthreads();
    if (det) {
        for (int q = threadIdx.x; q < NSLOT * NM * NM; q += blockDim.x) {
            double v = nbrAll[q];
            for (int w = 1; w < nWaves; ++w) v += nbrAll[w * (NSLOT * NM * NM) + q];
            nbrAll[q] = v;
        }
        __syncthreads();
        nbr = nbrAll;
    }
    if (stencil) {
        constexpr int CENTRE = NSLOT / 2;
        for (int q = threadIdx.x; q < NSLOT * NM * NM; q += blockDim.x) {
            const int slot = q / (NM * NM), e = q - slot * NM * NM;
            double v = nbr[q];
            if (slot == CENTRE) { v = 0; for (int w = 0; w < nWaves; ++w) v += diagRed[w * NM * NM + e]; }
            stencil[(int64_t)a * NSLOT * NM * NM + q] = v;
        }
        return;
    }
    // rows (a, .) of Ac belong to this workgroup: plain read-modify-write (the fallback atomics above may have touched them too)
    for (int q = threadIdx.x; q < NM * NM; q += blockDim.x) {
        double v = 0;
        for (int w = 0; w < nWaves; ++w) v += diagRed[w * NM * NM + q];
        unsafeAtomicAdd(&Ac[((int64_t)a * NM + q / NM) * m + (int64_t)a * NM + q % NM], v);
    }
    for (int q = threadIdx.x; q < NSLOT * NM * NM; q += blockDim.x) {
        const int slot = q / (NM * NM), e = q - slot * NM * NM;
        const int b = nbrAgg[slot];
        if (b < 0) continue;
        unsafeAtomicAdd(&Ac[((int64_t)a * NM + e / NM) * m + (int64_t)b * NM + e % NM], nbr[q]);
    }
}

// symmetrise the raw coarse operator into the padded matrix the dense inverse works on; modes without
// support (dead) are decoupled, the diagonal gets a tiny relative shift, the padding is scaled identity
__global__ void __launch_bounds__(256) k_tl_prep(int64_t m, int64_t mp, const double *__restrict__ Ac, const uint8_t *__restrict__ dead,
                                                 double maxd, double *__restrict__ Ap) {
    const int64_t total = mp * mp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / mp, j = e - i * mp;
        double v;
        if (i >= m || j >= m) v = (i == j) ? maxd : 0.0;
        else if (dead[i] || dead[j]) v = (i == j) ? maxd : 0.0;
        else {
            v = 0.5 * (Ac[i * m + j] + Ac[j * m + i]);
            if (i == j) v *= 1.0 + 1e-10;
        }
        Ap[e] = v;
    }
}

// y = A x for the dense coarse inverse (row-major m x m); one workgroup per row
__global__ void __launch_bounds__(256) k_tl_gemv(int64_t m, int64_t ld, const double *__restrict__ A, const double *__restrict__ x, double *__restrict__ y) {
    __shared__ double red[8];
    const int64_t row = blockIdx.x;
    double acc[1] = {0};
    for (int64_t j = threadIdx.x; j < m; j += 256) acc[0] += A[row * ld + j] * x[j];
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) y[row] = acc[0];
}

// z = D^-1 r + Z yc ; optionally accumulates r.z into *rzOut
template <int DIM>
__global__ void __launch_bounds__(256) k_tl_apply(TLArgs t, const double *__restrict__ dinv, const double *__restrict__ r,
                                                  const double *__restrict__ yc, double *__restrict__ z, double *scal, int it,
                                                  const double *stopPtr, DetBuf det) {
    __shared__ double red[8];
    double *rzOut = nullptr;
    if (scal) {
        it += (int)stopPtr[3];
        if (it >= 0 && scal[(int64_t)it * 4 + 2] <= stopPtr[0]) return;
        rzOut = scal + (int64_t)(it + 1) * 4;
    }
    double acc[1] = {0};
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < t.nDoF; n += (int64_t)gridDim.x * 256) {
        double rv[DIM], zv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) rv[c] = r[n * DIM + c];
        apply_block<DIM>(dinv + n * (DIM * (DIM + 1) / 2), rv, zv);
        const int a = t.aggOfDof[n];
        double rp[3] = {t.relPos[n * 3], t.relPos[n * 3 + 1], t.relPos[n * 3 + 2]};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (k >= t.nModes) break;
            const double y = yc[(int64_t)a * t.nModes + k];
#pragma unroll
            for (int c = 0; c < DIM; ++c) zv[c] += y * tl_mode<DIM>(k, c, rp);
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            if (t.fixedMask && t.fixedMask[n * DIM + c]) zv[c] = rv[c];   // identity on fixed variables (r is 0 there)
            z[n * DIM + c] = zv[c];
            acc[0] += rv[c] * zv[c];
        }
    }
    if (rzOut) {
        block_sum<1>(acc, red);
        double *const tg[1] = {rzOut};
        commit_sums<1>(acc, tg, det, red);
    }
}

// ------------------------------------------------------------------------------------------------
// Dense SPD inverse on the device for the coarse operator (m <= ~16k): blocked Cholesky with 64x64
// tiles -> L^-1 by block forward substitution -> A^-1 = L^-T L^-1. FP64, LDS-tiled 4x4 micro-tiles.
// The matrix is padded to a multiple of 64 with an identity block by the caller, so every tile is full.
// (rocSOLVER is deliberately not used: PyTorch wheels ship their own rocBLAS/rocSOLVER and mixing the
// two ROCm stacks in one process is not safe.)
// ------------------------------------------------------------------------------------------------
constexpr int DT = 64;         // tile edge
constexpr int DTP = DT + 1;    // LDS leading dimension (bank-conflict padding)

// C(64x64, registers 4x4 per thread) += A_s(64x64) * B_s(64x64), both in LDS as [row][DTP]
DEV void dense_tile_mma(const double *As, const double *Bs, double (&c)[4][4], int ty, int tx) {
#pragma unroll 4
    for (int q = 0; q < DT; ++q) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = As[(ty * 4 + i) * DTP + q];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[q * DTP + tx * 4 + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i][j] += a[i] * b[j];
    }
}
// load a 64x64 tile of a row-major matrix (leading dim ld) into LDS, optionally transposed
DEV void dense_tile_load(const double *__restrict__ G, int64_t ld, double *S, bool transpose) {
    for (int e = threadIdx.x; e < DT * DT; e += 256) {
        const int r = e / DT, c = e % DT;
        const double v = G[(int64_t)r * ld + c];
        if (transpose) S[c * DTP + r] = v; else S[r * DTP + c] = v;
    }
}

// diagonal tile: Cholesky in LDS, L_kk written back (upper zeroed), its inverse to Dinv; *notSpd set on failure
__global__ void __launch_bounds__(256) k_dense_potrf(double *A, int64_t ld, int k, double *Dinv, int *notSpd) {
    __shared__ double L[DT * DTP];
    __shared__ double X[DT * DTP];
    double *Akk = A + ((int64_t)k * DT) * ld + (int64_t)k * DT;
    dense_tile_load(Akk, ld, L, false);
    __syncthreads();
    for (int j = 0; j < DT; ++j) {
        if (threadIdx.x == 0) {
            const double d = L[j * DTP + j];
            if (!(d > 0)) { *notSpd = 1; L[j * DTP + j] = 1.0; } else L[j * DTP + j] = sqrt(d);
        }
        __syncthreads();
        const double djj = L[j * DTP + j];
        for (int i = j + 1 + threadIdx.x; i < DT; i += 256) L[i * DTP + j] /= djj;
        __syncthreads();
        // trailing update of the lower triangle: L[i][c] -= L[i][j] * L[c][j], j < c <= i
        const int rem = DT - j - 1;
        for (int e = threadIdx.x; e < rem * rem; e += 256) {
            const int i = j + 1 + e / rem, c = j + 1 + e % rem;
            if (c <= i) L[i * DTP + c] -= L[i * DTP + j] * L[c * DTP + j];
        }
        __syncthreads();
    }
    // X = L^-1 (lower): one thread per column
    if (threadIdx.x < DT) {
        const int c = threadIdx.x;
        for (int i = 0; i < DT; ++i) {
            if (i < c) { X[i * DTP + c] = 0.0; continue; }
            double sacc = (i == c) ? 1.0 : 0.0;
            for (int q = c; q < i; ++q) sacc -= L[i * DTP + q] * X[q * DTP + c];
            X[i * DTP + c] = sacc / L[i * DTP + i];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < DT * DT; e += 256) {
        const int r = e / DT, c = e % DT;
        Akk[(int64_t)r * ld + c] = c <= r ? L[r * DTP + c] : 0.0;
        Dinv[(int64_t)k * DT * DT + e] = X[r * DTP + c];
    }
}

// panel: L_ik = A_ik * Linv_kk^T for i > k
__global__ void __launch_bounds__(256) k_dense_trsm(double *A, int64_t ld, int k, const double *Dinv) {
    __shared__ double As[DT * DTP];
    __shared__ double Bs[DT * DTP];
    const int i = k + 1 + blockIdx.x;
    double *Aik = A + ((int64_t)i * DT) * ld + (int64_t)k * DT;
    dense_tile_load(Aik, ld, As, false);
    dense_tile_load(Dinv + (int64_t)k * DT * DT, DT, Bs, true);     // Bs[q][c] = Linv[c][q]
    __syncthreads();
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    double c[4][4] = {};
    dense_tile_mma(As, Bs, c, ty, tx);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) Aik[(int64_t)(ty * 4 + a) * ld + tx * 4 + b] = c[a][b];
}

// trailing update: A_ij -= L_ik L_jk^T for k < j <= i ; 2D grid over (i-k-1, j-k-1)
__global__ void __launch_bounds__(256) k_dense_syrk(double *A, int64_t ld, int k) {
    const int i = k + 1 + blockIdx.y, j = k + 1 + blockIdx.x;
    if (j > i) return;
    __shared__ double As[DT * DTP];
    __shared__ double Bs[DT * DTP];
    dense_tile_load(A + ((int64_t)i * DT) * ld + (int64_t)k * DT, ld, As, false);
    dense_tile_load(A + ((int64_t)j * DT) * ld + (int64_t)k * DT, ld, Bs, true);   // Bs[q][c] = L_jk[c][q]
    __syncthreads();
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    double c[4][4] = {};
    dense_tile_mma(As, Bs, c, ty, tx);
    double *Aij = A + ((int64_t)i * DT) * ld + (int64_t)j * DT;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) Aij[(int64_t)(ty * 4 + a) * ld + tx * 4 + b] -= c[a][b];
}

// X = L^-1, sub-diagonal d: X_{c+d,c} = -Linv_{c+d} * sum_{q=c}^{c+d-1} L_{c+d,q} X_{q,c}; d = 0 copies Linv
__global__ void __launch_bounds__(256) k_dense_trinv(const double *A, double *X, int64_t ld, int d, const double *Dinv) {
    __shared__ double As[DT * DTP];
    __shared__ double Bs[DT * DTP];
    const int cblk = blockIdx.x, i = cblk + d;
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    double *Xic = X + ((int64_t)i * DT) * ld + (int64_t)cblk * DT;
    if (d == 0) {
        for (int e = threadIdx.x; e < DT * DT; e += 256) Xic[(int64_t)(e / DT) * ld + e % DT] = Dinv[(int64_t)i * DT * DT + e];
        return;
    }
    double s[4][4] = {};
    for (int q = cblk; q < i; ++q) {
        __syncthreads();
        dense_tile_load(A + ((int64_t)i * DT) * ld + (int64_t)q * DT, ld, As, false);
        dense_tile_load(X + ((int64_t)q * DT) * ld + (int64_t)cblk * DT, ld, Bs, false);
        __syncthreads();
        dense_tile_mma(As, Bs, s, ty, tx);
    }
    __syncthreads();
    // As <- Linv_ii, Bs <- S ; X_ic = -Linv_ii * S
    dense_tile_load(Dinv + (int64_t)i * DT * DT, DT, As, false);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) Bs[(ty * 4 + a) * DTP + tx * 4 + b] = s[a][b];
    __syncthreads();
    double c[4][4] = {};
    dense_tile_mma(As, Bs, c, ty, tx);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) Xic[(int64_t)(ty * 4 + a) * ld + tx * 4 + b] = -c[a][b];
}

// Ainv_IJ = sum_{Q >= I} X_QI^T X_QJ for J <= I (and its mirror); 2D grid (J, I)
// L^-1 by recursive doubling instead of 94 dependent diagonal sweeps: with the inverses X11, X22 of two adjacent
// diagonal blocks of bt tiles known, the block below the diagonal is X21 = -X22 (L21 X11): two batched tile GEMMs per
// level, log2(nt) levels, every tile of a level independent.   step 1: W = L21 X11 (W in scratch), step 2: X21 = -X22 W.
__global__ void __launch_bounds__(256) k_dense_linv_level(const double *__restrict__ L, double *X, double *W, int64_t ld, int nt, int bt, int step) {
    __shared__ double As[DT * DTP];
    __shared__ double Bs[DT * DTP];
    const int pair = blockIdx.z, c0 = 2 * pair * bt, r0 = c0 + bt;
    const int I = blockIdx.y, J = blockIdx.x;
    if (r0 + I >= nt || r0 + I >= r0 + bt) return;
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    double c[4][4] = {};
    if (step == 1) {
        for (int q = J; q < bt; ++q) {            // X11 is lower triangular: tiles (q, J) with q >= J
            __syncthreads();
            dense_tile_load(L + ((int64_t)(r0 + I) * DT) * ld + (int64_t)(c0 + q) * DT, ld, As, false);
            dense_tile_load(X + ((int64_t)(c0 + q) * DT) * ld + (int64_t)(c0 + J) * DT, ld, Bs, false);
            __syncthreads();
            dense_tile_mma(As, Bs, c, ty, tx);
        }
        double *o = W + ((int64_t)(r0 + I) * DT) * ld + (int64_t)(c0 + J) * DT;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) o[(int64_t)(ty * 4 + a) * ld + tx * 4 + b] = c[a][b];
    } else {
        for (int q = 0; q <= I; ++q) {            // X22 is lower triangular: tiles (I, q) with q <= I
            __syncthreads();
            dense_tile_load(X + ((int64_t)(r0 + I) * DT) * ld + (int64_t)(r0 + q) * DT, ld, As, false);
            dense_tile_load(W + ((int64_t)(r0 + q) * DT) * ld + (int64_t)(c0 + J) * DT, ld, Bs, false);
            __syncthreads();
            dense_tile_mma(As, Bs, c, ty, tx);
        }
        double *o = X + ((int64_t)(r0 + I) * DT) * ld + (int64_t)(c0 + J) * DT;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) o[(int64_t)(ty * 4 + a) * ld + tx * 4 + b] = -c[a][b];
    }
}
__global__ void __launch_bounds__(256) k_dense_xtx(const double *X, double *Ainv, int64_t ld, int nt) {
    const int I = blockIdx.y, J = blockIdx.x;
    if (J > I) return;
    __shared__ double As[DT * DTP];
    __shared__ double Bs[DT * DTP];
    const int ty = threadIdx.x / 16, tx = threadIdx.x % 16;
    double c[4][4] = {};
    for (int Q = I; Q < nt; ++Q) {
        __syncthreads();
        dense_tile_load(X + ((int64_t)Q * DT) * ld + (int64_t)I * DT, ld, As, true);    // As[r][q] = X_QI[q][r]
        dense_tile_load(X + ((int64_t)Q * DT) * ld + (int64_t)J * DT, ld, Bs, false);
        __syncthreads();
        dense_tile_mma(As, Bs, c, ty, tx);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = ty * 4 + a, cc = tx * 4 + b;
            Ainv[((int64_t)I * DT + r) * ld + (int64_t)J * DT + cc] = c[a][b];
            Ainv[((int64_t)J * DT + cc) * ld + (int64_t)I * DT + r] = c[a][b];
        }
}

// ------------------------------------------------------------------------------------------------
// PCG vector kernels. scal[it*4 + {0: r.z, 1: p.Ap, 2: r.r}] hold the reductions of iteration `it`
// (array zero-filled once per solve; nothing is reset inside the loop).
// ------------------------------------------------------------------------------------------------
template <int DIM>
__global__ void __launch_bounds__(256) k_pcg_init(int64_t nRows, const double *__restrict__ dinv, const double *__restrict__ b,
                                                  double *__restrict__ x, double *__restrict__ r, double *__restrict__ z,
                                                  double *__restrict__ p, double *scal, DetBuf det) {
    __shared__ double red[16];
    double acc[2] = {0, 0};
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nRows; n += (int64_t)gridDim.x * 256) {
        double rv[DIM], zv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) rv[c] = b[n * DIM + c];
        apply_block<DIM>(dinv + n * (DIM * (DIM + 1) / 2), rv, zv);
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            x[n * DIM + c] = 0.0; r[n * DIM + c] = rv[c]; z[n * DIM + c] = zv[c]; p[n * DIM + c] = zv[c];
            acc[0] += rv[c] * zv[c]; acc[1] += rv[c] * rv[c];
        }
    }
    block_sum<2>(acc, red);
    double *const tg[2] = {&scal[0], &scal[2]};
    commit_sums<2>(acc, tg, det, red);
}

// r -= alpha Ap ; z = Dinv r ; scal[it+1].{rz,rr} += ...   (x += alpha p happens in k_pcg_direction, which reads p anyway: one
// vector read less per iteration than updating x here)
// ZS (the multigrid preconditioner's pre-smoothing from zero folded in, mg_precond's MgFuse): z = zs Dinv r, and r.z is NOT accumulated (the
// preconditioner is not finished: the V-cycle's last kernel forms it, k_mg_cheb_rz)
template <int DIM, bool SKIPZ = false, bool ZS = false, class DinvT = double>
__global__ void __launch_bounds__(256) k_pcg_update(int64_t nRows, const DinvT *__restrict__ dinv,
                                                    const double *__restrict__ Ap, double *__restrict__ r,
                                                    double *__restrict__ z, double *scal, int it, const double *stopPtr, DetBuf det, double zs = 1.0) {
    __shared__ double red[16];
    it += (int)stopPtr[3];
    if (scal[(int64_t)it * 4 + 2] <= stopPtr[0]) return;
    const double alpha = scal[(int64_t)it * 4 + 0] / scal[(int64_t)it * 4 + 1];
    double acc[2] = {0, 0};
    constexpr int NS = DIM * (DIM + 1) / 2;
    // a lane takes TWO consecutive rows: 2 DIM doubles of Ap / r / z and 2 NS of the packed inverse blocks are contiguous and
    // 16-byte aligned, so everything moves in 16-byte accesses (the arrays come from hipMalloc)
    const int64_t nPair = nRows >> 1;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nPair; q += (int64_t)gridDim.x * 256) {
        double av[2 * DIM], rv[2 * DIM], zv[2 * DIM], dm[2 * NS];
        const double2 *a2 = reinterpret_cast<const double2 *>(Ap + q * 2 * DIM);
        double2 *r2 = reinterpret_cast<double2 *>(r + q * 2 * DIM);
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const double2 t = a2[c], u = r2[c];
            av[2 * c] = t.x; av[2 * c + 1] = t.y; rv[2 * c] = u.x; rv[2 * c + 1] = u.y;
        }
#pragma unroll
        for (int c = 0; c < 2 * DIM; ++c) rv[c] -= alpha * av[c];
#pragma unroll
        for (int c = 0; c < DIM; ++c) r2[c] = make_double2(rv[2 * c], rv[2 * c + 1]);
        if (!SKIPZ) {
            load_dinv_pair<NS>(dinv + q * 2 * NS, dm);
            apply_block<DIM>(dm, rv, zv);
            apply_block<DIM>(dm + NS, rv + DIM, zv + DIM);
            double2 *z2 = reinterpret_cast<double2 *>(z + q * 2 * DIM);
            if (ZS)
#pragma unroll
                for (int c = 0; c < 2 * DIM; ++c) zv[c] *= zs;
#pragma unroll
            for (int c = 0; c < DIM; ++c) z2[c] = make_double2(zv[2 * c], zv[2 * c + 1]);
            if (!ZS)
#pragma unroll
                for (int c = 0; c < 2 * DIM; ++c) acc[0] += rv[c] * zv[c];
        }
#pragma unroll
        for (int c = 0; c < 2 * DIM; ++c) acc[1] += rv[c] * rv[c];
    }
    if ((nRows & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // the odd last row
        const int64_t n = nRows - 1;
        double rv[DIM], zv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const int64_t g = n * DIM + c;
            rv[c] = r[g] - alpha * Ap[g];
            r[g] = rv[c];
        }
        if (!SKIPZ) {
            apply_block_t<DIM, DinvT>(dinv + n * NS, rv, zv);
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                if (ZS) z[n * DIM + c] = zs * zv[c];
                else { z[n * DIM + c] = zv[c]; acc[0] += rv[c] * zv[c]; }
            }
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) acc[1] += rv[c] * rv[c];
    }
    block_sum<2>(acc, red);
    double *const tg[2] = {(SKIPZ || ZS) ? nullptr : &scal[(int64_t)(it + 1) * 4 + 0], &scal[(int64_t)(it + 1) * 4 + 2]};
    commit_sums<2>(acc, tg, det, red);
}

thread_local DetBuf t_det;

// second stage of commit_sums: one workgroup adds the producers' partials in workgroup order (thread t takes b = t, t + 256, ...; then the
// fixed tree of block_sum) and adds the totals onto the targets the producer named
__global__ void __launch_bounds__(256) k_det_finish(double *scratch) {
    __shared__ double red[20];
    unsigned long long *h = reinterpret_cast<unsigned long long *>(scratch);
    const int nv = (int)h[0];
    if (nv == 0) return;
    const int64_t nb = (int64_t)h[1];
    double s4[4] = {0, 0, 0, 0};
    for (int64_t b = threadIdx.x; b < nb; b += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nv) s4[k] += scratch[DET_HEADER + b * 4 + k];
    block_sum<4>(s4, red);
    if (threadIdx.x == 0) {
        for (int k = 0; k < nv; ++k) {
            double *t = reinterpret_cast<double *>(h[2 + k]);
            if (t) *t += s4[k];
        }
        h[0] = 0;
    }
}
void launch_det_finish(hipStream_t s) {
    if (!t_det.partials) return;
    hipLaunchKernelGGL(k_det_finish, dim3(1), dim3(256), 0, s, t_det.partials);
}
int g_vecGridCap = 16384;   // workgroups of k_pcg_direction (option "vec_grid_cap"; 0.899 vs 0.911 ms per iteration against 2048). Kernels that end in a
                            // reduction keep 2048: 16384 workgroups x 2 atomics on the same scalars cost 0.14 ms
// x += alpha p ; p = z + beta p   (alpha of this iteration, beta from the reductions k_pcg_update / the preconditioner just made)
__global__ void __launch_bounds__(256) k_pcg_direction(int64_t n, const double *__restrict__ z, double *__restrict__ p, double *__restrict__ x,
                                                       const double *scal, int it, const double *stopPtr) {
    it += (int)stopPtr[3];
    if (scal[(int64_t)it * 4 + 2] <= stopPtr[0]) return;
    const double alpha = scal[(int64_t)it * 4 + 0] / scal[(int64_t)it * 4 + 1];
    const double beta = scal[(int64_t)(it + 1) * 4 + 0] / scal[(int64_t)it * 4 + 0];
    // 16 bytes per lane and access (the arrays come from hipMalloc: 256-byte aligned)
    const int64_t n2 = n >> 1;
    const double2 *z2 = reinterpret_cast<const double2 *>(z);
    double2 *p2 = reinterpret_cast<double2 *>(p), *x2 = reinterpret_cast<double2 *>(x);
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n2; k += (int64_t)gridDim.x * 256) {
        const double2 pv = p2[k], zv = z2[k];
        double2 xv = x2[k];
        xv.x += alpha * pv.x; xv.y += alpha * pv.y;
        x2[k] = xv;
        p2[k] = make_double2(zv.x + beta * pv.x, zv.y + beta * pv.y);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const double pv = p[n - 1];
        x[n - 1] += alpha * pv;
        p[n - 1] = z[n - 1] + beta * pv;
    }
}

// distributed PCG building blocks: the scalars live in device memory (results of RCCL all-reduces), so no host sync
// x += a p ; r -= a Ap  with a = num[0] / den[0]
__global__ void __launch_bounds__(256) k_dev_update_xr(int64_t n, const double *num, const double *den, const double *__restrict__ p,
                                                       const double *__restrict__ Ap, double *__restrict__ x, double *__restrict__ r) {
    const double a = num[0] / den[0];
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        x[k] += a * p[k];
        r[k] -= a * Ap[k];
    }
}
// p = z + b p  with b = num[0] / den[0]
__global__ void __launch_bounds__(256) k_dev_direction(int64_t n, const double *num, const double *den, const double *__restrict__ z,
                                                       double *__restrict__ p) {
    const double b = num[0] / den[0];
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) p[k] = z[k] + b * p[k];
}
// out[0] = r.z, out[1] = r.r (out zeroed by the caller)
__global__ void __launch_bounds__(256) k_dev_dots(int64_t n, const double *__restrict__ r, const double *__restrict__ z, double *out, DetBuf det) {
    __shared__ double red[16];
    double acc[2] = {0, 0};
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) { acc[0] += r[k] * z[k]; acc[1] += r[k] * r[k]; }
    block_sum<2>(acc, red);
    double *const tg[2] = {&out[0], &out[1]};
    commit_sums<2>(acc, tg, det, red);
}

// stop[3] += n: advances the iteration base at the end of a captured block of PCG iterations
__global__ void k_advance_base(double *stop, double n) { stop[3] += n; }
__global__ void k_add_scalar(double *p, double v) { *p += v; }

__global__ void __launch_bounds__(256) k_axpby(int64_t n, double a, const double *__restrict__ x, double b, double *__restrict__ y) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        y[k] = a * x[k] + (b == 0.0 ? 0.0 : b * y[k]);
}
__global__ void __launch_bounds__(256) k_mask(int64_t n, const uint8_t *__restrict__ m, double *__restrict__ v) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        if (m[k]) v[k] = 0.0;
}
// v[idx[k]] = val[k] for the indices below `bound` (the length of v: a vector of the owned rows of a partitioned context
// must not receive the fixed values of halo variables, whose indices lie past its end)
__global__ void __launch_bounds__(256) k_scatter_values(int64_t n, const int64_t *__restrict__ idx, const double *__restrict__ val,
                                                        double *__restrict__ v, int64_t bound) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        if (idx[k] < bound) v[idx[k]] = val[k];
}
__global__ void __launch_bounds__(256) k_dot(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *out, DetBuf det) {
    __shared__ double red[8];
    double acc[1] = {0};
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) acc[0] += a[k] * b[k];
    block_sum<1>(acc, red);
    double *const tg[1] = {out};
    commit_sums<1>(acc, tg, det, red);
}
// tiled -> array-of-blocks (export)
__global__ void __launch_bounds__(256) k_untile(int NB, int64_t nnzb, const double *__restrict__ tiled, double *__restrict__ aos) {
    const int64_t total = nnzb * NB;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t s = k / NB;
        const int c = (int)(k - s * NB);
        aos[k] = tiled[tiled_index(s, c, NB)];
    }
}


// ------------------------------------------------------------------------------------------------
// Chronopoulos-Gear PCG (mfh_solver.cpp): ONE reduction point per iteration, NR interleaved right-hand sides.
//   scal[(it NR + k) 4 + {0: gamma = (r, u), 1: delta = (w, u), 2: rr = (r, r), 3: alpha}],  u = M^-1 r,  w = K u
//   ctl[0] = iteration base (graph replay), ctl[2 + k] = rtol^2 (b_k, b_k)
//   beta = gamma_it / gamma_it-1, alpha = gamma_it / (delta_it - beta gamma_it / alpha_it-1)
//   p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s ; u = D^-1 r ; gamma_it+1, rr_it+1
// A converged right-hand side (rr_k(it) <= ctl[2 + k]) is frozen: its vectors are left alone and it adds nothing to
// scal[it + 1], so rr_k stays 0 <= stop from then on. On a row-partitioned context scal[it + 1] holds this rank's partial
// sums until the solver all-reduces it (alpha is written afterwards, by this kernel).
// ------------------------------------------------------------------------------------------------
// One lane per (row, vector) pair g = row * NR + k: consecutive lanes touch consecutive DIM-vectors of every array whatever
// NR is (a lane per ROW would stride by NR * DIM doubles: measured 19x slower at NR = 6). The launch uses a grid that is
// a multiple of 3 workgroups, so that the total lane count is a multiple of every supported NR and a lane keeps its k.
template <int DIM, bool SKIPU>
__global__ void __launch_bounds__(256) k_cg_update(int64_t nRows, int NR, const double *__restrict__ dinv, double *__restrict__ u,
                                                   const double *__restrict__ w, double *__restrict__ p, double *__restrict__ sv,
                                                   double *__restrict__ x, double *__restrict__ r, double *scal, int it, const double *ctl) {
    __shared__ double sAl[8], sBe[8], sAcc[16];
    __shared__ int sDn[8], sAll;
    it += (int)ctl[0];
    if (threadIdx.x < 16) sAcc[threadIdx.x] = 0.0;
    if (threadIdx.x == 0) {
        int all = 1;
        for (int k = 0; k < NR; ++k) {
            const double *sc = scal + ((int64_t)it * NR + k) * 4;
            const int dn = sc[2] <= ctl[2 + k];
            all &= dn;
            double be = 0.0, al = sc[0] / sc[1];
            if (it > 0) {
                const double gp = sc[-4 * NR], ap = sc[-4 * NR + 3];
                be = sc[0] / gp;
                al = sc[0] / (sc[1] - be * sc[0] / ap);
            }
            sAl[k] = al; sBe[k] = be; sDn[k] = dn;
            if (blockIdx.x == 0 && !dn) scal[((int64_t)it * NR + k) * 4 + 3] = al;
        }
        sAll = all;
    }
    __syncthreads();
    if (sAll) return;
    const int64_t nPairs = nRows * NR, stride = (int64_t)gridDim.x * 256;
    const int64_t g0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k = (int)(g0 % NR);
    const bool dn = sDn[k] != 0;
    const double al = sAl[k], be = sBe[k];
    double accG = 0.0, accR = 0.0;
    const int64_t rowStride = stride / NR;   // the lane count is a multiple of NR: a lane keeps its k, its row advances by this
    int64_t n = g0 / NR;
    if (!dn)
        for (int64_t g = g0; g < nPairs; g += stride, n += rowStride) {
            double rv[DIM], zv[DIM];
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                const int64_t q = g * DIM + c;
                const double pv = u[q] + be * p[q];
                const double sn = w[q] + be * sv[q];
                p[q] = pv;
                sv[q] = sn;
                x[q] += al * pv;
                rv[c] = r[q] - al * sn;
                r[q] = rv[c];
            }
            if (!SKIPU) {
                apply_block<DIM>(dinv + n * (DIM * (DIM + 1) / 2), rv, zv);
#pragma unroll
                for (int c = 0; c < DIM; ++c) { u[g * DIM + c] = zv[c]; accG += rv[c] * zv[c]; }
            }
#pragma unroll
            for (int c = 0; c < DIM; ++c) accR += rv[c] * rv[c];
        }
    if (!dn) {
        if (!SKIPU) unsafeAtomicAdd(&sAcc[2 * k], accG);
        unsafeAtomicAdd(&sAcc[2 * k + 1], accR);
    }
    __syncthreads();
    if (threadIdx.x < 2 * NR && !sDn[threadIdx.x >> 1]) {
        const int kk = threadIdx.x >> 1, which = threadIdx.x & 1;
        if (!(SKIPU && which == 0)) unsafeAtomicAdd(&scal[((int64_t)(it + 1) * NR + kk) * 4 + (which ? 2 : 0)], sAcc[threadIdx.x]);
    }
}

// The same for ONE right-hand side (the distributed solve's usual case): two consecutive rows per lane, every array in
// 16-byte accesses (like k_pcg_update).
template <int DIM, bool SKIPU>
__global__ void __launch_bounds__(256) k_cg_update1(int64_t nRows, const double *__restrict__ dinv, double *__restrict__ u,
                                                    const double *__restrict__ w, double *__restrict__ p, double *__restrict__ sv,
                                                    double *__restrict__ x, double *__restrict__ r, double *scal, int it, const double *ctl) {
    __shared__ double red[16];
    constexpr int NS = DIM * (DIM + 1) / 2;
    it += (int)ctl[0];
    const double *sc = scal + (int64_t)it * 4;
    if (sc[2] <= ctl[2]) return;
    double be = 0.0, al = sc[0] / sc[1];
    if (it > 0) {
        const double gp = sc[-4], ap = sc[-4 + 3];
        be = sc[0] / gp;
        al = sc[0] / (sc[1] - be * sc[0] / ap);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[(int64_t)it * 4 + 3] = al;
    double acc[2] = {0, 0};
    const int64_t nPair = nRows >> 1;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nPair; q += (int64_t)gridDim.x * 256) {
        double pv[2 * DIM], sn[2 * DIM], xv[2 * DIM], rv[2 * DIM], zv[2 * DIM];
        const int64_t base = q * 2 * DIM;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const double2 a = reinterpret_cast<const double2 *>(u + base)[c], b = reinterpret_cast<const double2 *>(w + base)[c],
                          pp = reinterpret_cast<const double2 *>(p + base)[c], ss = reinterpret_cast<const double2 *>(sv + base)[c],
                          xx = reinterpret_cast<const double2 *>(x + base)[c], rr = reinterpret_cast<const double2 *>(r + base)[c];
            pv[2 * c] = a.x + be * pp.x; pv[2 * c + 1] = a.y + be * pp.y;
            sn[2 * c] = b.x + be * ss.x; sn[2 * c + 1] = b.y + be * ss.y;
            xv[2 * c] = xx.x + al * pv[2 * c]; xv[2 * c + 1] = xx.y + al * pv[2 * c + 1];
            rv[2 * c] = rr.x - al * sn[2 * c]; rv[2 * c + 1] = rr.y - al * sn[2 * c + 1];
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            reinterpret_cast<double2 *>(p + base)[c] = make_double2(pv[2 * c], pv[2 * c + 1]);
            reinterpret_cast<double2 *>(sv + base)[c] = make_double2(sn[2 * c], sn[2 * c + 1]);
            reinterpret_cast<double2 *>(x + base)[c] = make_double2(xv[2 * c], xv[2 * c + 1]);
            reinterpret_cast<double2 *>(r + base)[c] = make_double2(rv[2 * c], rv[2 * c + 1]);
        }
        if (!SKIPU) {
            double dm[2 * NS];
            const double2 *d2 = reinterpret_cast<const double2 *>(dinv + q * 2 * NS);
#pragma unroll
            for (int c = 0; c < NS; ++c) { const double2 t = d2[c]; dm[2 * c] = t.x; dm[2 * c + 1] = t.y; }
            apply_block<DIM>(dm, rv, zv);
            apply_block<DIM>(dm + NS, rv + DIM, zv + DIM);
#pragma unroll
            for (int c = 0; c < DIM; ++c) reinterpret_cast<double2 *>(u + base)[c] = make_double2(zv[2 * c], zv[2 * c + 1]);
#pragma unroll
            for (int c = 0; c < 2 * DIM; ++c) acc[0] += rv[c] * zv[c];
        }
#pragma unroll
        for (int c = 0; c < 2 * DIM; ++c) acc[1] += rv[c] * rv[c];
    }
    if ((nRows & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // the odd last row
        const int64_t n = nRows - 1;
        double rv[DIM], zv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const int64_t g = n * DIM + c;
            const double pv = u[g] + be * p[g], sn = w[g] + be * sv[g];
            p[g] = pv; sv[g] = sn; x[g] += al * pv;
            rv[c] = r[g] - al * sn;
            r[g] = rv[c];
        }
        if (!SKIPU) {
            apply_block<DIM>(dinv + n * NS, rv, zv);
#pragma unroll
            for (int c = 0; c < DIM; ++c) { u[n * DIM + c] = zv[c]; acc[0] += rv[c] * zv[c]; }
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) acc[1] += rv[c] * rv[c];
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        if (!SKIPU) unsafeAtomicAdd(&scal[(int64_t)(it + 1) * 4 + 0], acc[0]);
        unsafeAtomicAdd(&scal[(int64_t)(it + 1) * 4 + 2], acc[1]);
    }
}

// start of the solve: u = D^-1 r, gamma_0 = (r, u), rr_0 = (r, r) into scal[0]
template <int DIM, bool SKIPU>
__global__ void __launch_bounds__(256) k_cg_init(int64_t nRows, int NR, const double *__restrict__ dinv, const double *__restrict__ r,
                                                 double *__restrict__ u, double *scal) {
    __shared__ double sAcc[16];
    if (threadIdx.x < 16) sAcc[threadIdx.x] = 0.0;
    __syncthreads();
    const int64_t nPairs = nRows * NR, stride = (int64_t)gridDim.x * 256;
    const int64_t g0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int k = (int)(g0 % NR);
    double accG = 0.0, accR = 0.0;
    const int64_t rowStride = stride / NR;
    int64_t n = g0 / NR;
    for (int64_t g = g0; g < nPairs; g += stride, n += rowStride) {
        double rv[DIM], zv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) rv[c] = r[g * DIM + c];
        if (!SKIPU) {
            apply_block<DIM>(dinv + n * (DIM * (DIM + 1) / 2), rv, zv);
#pragma unroll
            for (int c = 0; c < DIM; ++c) { u[g * DIM + c] = zv[c]; accG += rv[c] * zv[c]; }
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) accR += rv[c] * rv[c];
    }
    if (!SKIPU) unsafeAtomicAdd(&sAcc[2 * k], accG);
    unsafeAtomicAdd(&sAcc[2 * k + 1], accR);
    __syncthreads();
    if (threadIdx.x < 2 * NR) {
        const int kk = threadIdx.x >> 1, which = threadIdx.x & 1;
        if (!(SKIPU && which == 0)) unsafeAtomicAdd(&scal[kk * 4 + (which ? 2 : 0)], sAcc[threadIdx.x]);
    }
}

// two-level preconditioner for NR interleaved vectors: coarse vectors are [coarse index][NR]
template <int DIM>
__global__ void __launch_bounds__(256) k_tl_restrict_nr(TLArgs t, int NR, const int32_t *__restrict__ aggPtr, const int32_t *__restrict__ dofsByAgg,
                                                        const double *__restrict__ w, double *__restrict__ rc) {
    __shared__ double red[4 * 6];
    const int a = blockIdx.x;
    const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
    for (int kr = 0; kr < NR; ++kr) {
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int q = aggPtr[a] + threadIdx.x; q < aggPtr[a + 1]; q += 256) {
            const int64_t n = dofsByAgg[q];
            double rp[3] = {t.relPos[n * 3], t.relPos[n * 3 + 1], t.relPos[n * 3 + 2]};
            double wv[DIM];
#pragma unroll
            for (int c = 0; c < DIM; ++c) wv[c] = (t.fixedMask && t.fixedMask[n * DIM + c]) ? 0.0 : w[(n * NR + kr) * DIM + c];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (k >= t.nModes) break;
                double s = 0;
#pragma unroll
                for (int c = 0; c < DIM; ++c) s += tl_mode<DIM>(k, c, rp) * wv[c];
                acc[k] += s;
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] = wave_sum(acc[k]);
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[wv_ * 6 + k] = acc[k];
        __syncthreads();
        if (threadIdx.x < t.nModes)
            rc[((int64_t)a * t.nModes + threadIdx.x) * NR + kr] = red[threadIdx.x] + red[6 + threadIdx.x] + red[12 + threadIdx.x] + red[18 + threadIdx.x];
        __syncthreads();
    }
}

// Y[m][NR] = A[m][m] X[m][NR] for the dense coarse inverse: one workgroup per row, the row is read once for all NR
__global__ void __launch_bounds__(256) k_tl_gemv_nr(int64_t m, int64_t ld, int NR, const double *__restrict__ A, const double *__restrict__ x,
                                                    double *__restrict__ y) {
    __shared__ double red[4 * 6];
    const int64_t row = blockIdx.x;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t j = threadIdx.x; j < m; j += 256) {
        const double av = A[row * ld + j];
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < NR) acc[k] += av * x[j * NR + k];
    }
    block_sum<6>(acc, red);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < NR) y[row * NR + k] = acc[k];
}

// u = D^-1 r + Z yc for NR vectors (one lane per (row, vector) pair); accumulates gamma_k = (r_k, u_k) into
// scal[((it + 1) NR + k) 4] (it = -1: the start)
template <int DIM>
__global__ void __launch_bounds__(256) k_tl_apply_nr(TLArgs t, int NR, const double *__restrict__ dinv, const double *__restrict__ r,
                                                     const double *__restrict__ yc, double *__restrict__ z, double *scal, int it,
                                                     const double *ctl) {
    __shared__ double sAcc[8];
    __shared__ int sDn[8], sAll;
    if (threadIdx.x < 8) { sAcc[threadIdx.x] = 0.0; sDn[threadIdx.x] = 0; }
    if (threadIdx.x == 0) sAll = 0;
    __syncthreads();
    if (scal) {
        it += (int)ctl[0];
        if (it >= 0 && threadIdx.x == 0) {
            int all = 1;
            for (int k = 0; k < NR; ++k) { sDn[k] = scal[((int64_t)it * NR + k) * 4 + 2] <= ctl[2 + k]; all &= sDn[k]; }
            sAll = all;
        }
        __syncthreads();
        if (sAll) return;
    }
    const int64_t nPairs = t.nDoF * NR, stride = (int64_t)gridDim.x * 256;
    const int64_t g0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int kr = (int)(g0 % NR);
    const bool dn = sDn[kr] != 0;
    double acc = 0.0;
    const int64_t rowStride = stride / NR;
    int64_t n = g0 / NR;
    if (!dn)
        for (int64_t g = g0; g < nPairs; g += stride, n += rowStride) {
            const int a = t.aggOfDof[n];
            const double rp[3] = {t.relPos[n * 3], t.relPos[n * 3 + 1], t.relPos[n * 3 + 2]};
            double rv[DIM], zv[DIM];
#pragma unroll
            for (int c = 0; c < DIM; ++c) rv[c] = r[g * DIM + c];
            apply_block<DIM>(dinv + n * (DIM * (DIM + 1) / 2), rv, zv);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                if (k >= t.nModes) break;
                const double yv = yc[((int64_t)a * t.nModes + k) * NR + kr];
#pragma unroll
                for (int c = 0; c < DIM; ++c) zv[c] += yv * tl_mode<DIM>(k, c, rp);
            }
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                if (t.fixedMask && t.fixedMask[n * DIM + c]) zv[c] = rv[c];   // identity on fixed variables (r is 0 there)
                z[g * DIM + c] = zv[c];
                acc += rv[c] * zv[c];
            }
        }
    if (scal) {
        if (!dn) unsafeAtomicAdd(&sAcc[kr], acc);
        __syncthreads();
        if (threadIdx.x < NR && !sDn[threadIdx.x]) unsafeAtomicAdd(&scal[((int64_t)(it + 1) * NR + threadIdx.x) * 4], sAcc[threadIdx.x]);
    }
}

// halo send buffers: dst[j][:] = src[idx[j]][:], W doubles per row
__global__ void __launch_bounds__(256) k_pack_rows(int64_t n, int W, const int32_t *__restrict__ idx, const double *__restrict__ src,
                                                   double *__restrict__ dst) {
    const int64_t total = n * W;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t j = k / W;
        dst[k] = src[(int64_t)idx[j] * W + (k - j * W)];
    }
}
// the reverse of a halo exchange: dst[idx[j]][:] += src[j][:] (the entries of one peer's list are distinct rows; the peers are added one after
// the other, in peer order: the same sum on every run)
__global__ void __launch_bounds__(256) k_unpack_add_rows(int64_t n, int W, const int32_t *__restrict__ idx, const double *__restrict__ src,
                                                         double *__restrict__ dst) {
    const int64_t total = n * W;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t j = k / W;
        dst[(int64_t)idx[j] * W + (k - j * W)] += src[k];
    }
}
__global__ void __launch_bounds__(256) k_pack_rows_f32(int64_t n, int W, const int32_t *__restrict__ idx, const float *__restrict__ src,
                                                       float *__restrict__ dst) {
    const int64_t total = n * W;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t j = k / W;
        dst[k] = src[(int64_t)idx[j] * W + (k - j * W)];
    }
}
// v[k] = map[v[k]] (negative entries stay)
__global__ void __launch_bounds__(256) k_remap_i32(int64_t n, const int32_t *__restrict__ map, int32_t *__restrict__ v) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        const int32_t a = v[k];
        v[k] = a >= 0 ? map[a] : a;
    }
}
// [k][row][d] (separate vectors, the ABI's layout) <-> [row][k][d] (interleaved, the solver's layout)
__global__ void __launch_bounds__(256) k_interleave(int64_t nRows, int NR, int DIM, const double *__restrict__ src, double *__restrict__ dst,
                                                    int toInterleaved, int64_t srcStride) {
    const int64_t total = nRows * NR * DIM;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        const int64_t n = g / (NR * DIM);
        const int kc = (int)(g - n * (NR * DIM));
        const int k = kc / DIM, c = kc - k * DIM;
        const int64_t sep = (int64_t)k * srcStride + n * DIM + c;
        if (toInterleaved) dst[g] = src[sep]; else dst[sep] = src[g];
    }
}
// per-vector squared norms of NR interleaved vectors: out[k] += sum v_k^2
__global__ void __launch_bounds__(256) k_norms_nr(int64_t nRows, int NR, int DIM, const double *__restrict__ v, double *out) {
    __shared__ double red[4 * 6];
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nRows; n += (int64_t)gridDim.x * 256)
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < NR)
                for (int c = 0; c < DIM; ++c) { const double q = v[(n * NR + k) * DIM + c]; acc[k] += q * q; }
    block_sum<6>(acc, red);
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if (k < NR) unsafeAtomicAdd(&out[k], acc[k]);
}
// v[row][k][c] = 0 where the (row, c) variable is fixed
__global__ void __launch_bounds__(256) k_mask_nr(int64_t nRows, int NR, int DIM, const uint8_t *__restrict__ m, double *__restrict__ v) {
    const int64_t total = nRows * NR * DIM;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (int64_t)gridDim.x * 256) {
        const int64_t n = g / (NR * DIM);
        const int c = (int)(g % DIM);
        if (m[n * DIM + c]) v[g] = 0.0;
    }
}
// v[var(idx[j])][k] = val[j] for every k (fixed values are the same for all right-hand sides)
__global__ void __launch_bounds__(256) k_scatter_values_nr(int64_t n, int NR, int DIM, const int64_t *__restrict__ idx, const double *__restrict__ val,
                                                           double *__restrict__ v, int64_t rowBound) {
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        const int64_t row = idx[j] / DIM;
        if (row >= rowBound) continue;     // v holds rowBound rows (owned rows only, or owned + halo)
        const int c = (int)(idx[j] - row * DIM);
        for (int k = 0; k < NR; ++k) v[(row * NR + k) * DIM + c] = val[j];
    }
}

// ------------------------------------------------------------------------------------------------
// p-multigrid preconditioner (mfh_multigrid.cpp): quadratic level (matrix-free operator) -> linear level on the same
// vertices (assembled K1 = P^T K2 P exactly: P1 is a subspace of P2 and both quadratures are exact for their integrands)
// -> rigid-body modes of geometric aggregates (the dense coarse inverse of the two-level preconditioner). The prolongation is
// the embedding of the shape functions: a vertex DoF copies its coarse value, an edge-node DoF takes the mean of its two ends
// (phi^P1_v = phi^P2_v + 1/2 sum_{edges e at v} phi^P2_e; Functions.hh:238-318 nodal bases).
// Every kernel takes the PCG gate (scal, it, stop): a no-op once the residual of iteration it has met the threshold, like
// the vector kernels of the loop (blocks of check_every iterations are enqueued / replayed from a hipGraph without host sync).
// ------------------------------------------------------------------------------------------------
// nr == 0: the gate of ONE classic PCG loop (scal[it 4 + 2] against stop[0], iteration base stop[3]). nr >= 1: the batched V-cycle's coarse
// levels serve nr classic loops that advance in lockstep -- loop k keeps its history at scal + k stride and its control block at stop + 4 k --
// and are closed once EVERY one of them has converged.
struct MgGate { const double *scal; int it; const double *stop; int nr; int64_t stride; };
DEV bool mg_closed(const MgGate &g) {
    if (!g.scal) return false;
    if (g.nr > 0) {
        bool all = true;
        for (int k = 0; k < g.nr; ++k) {
            const double *st = g.stop + 4 * k;
            const int it = g.it + (int)st[3];
            all = all && it >= 0 && g.scal[k * g.stride + (int64_t)it * 4 + 2] <= st[0];
        }
        return all;
    }
    const int it = g.it + (int)g.stop[3];
    return it >= 0 && g.scal[(int64_t)it * 4 + 2] <= g.stop[0];
}

// One Chebyshev step on the block-Jacobi-preconditioned operator:  r' = rin - t ; d = a d + b D^-1 r' ; x (+)= d
//   t == null: r' = rin;  rout != null: the running residual is stored;  first: d is not read (a d = 0);  assign: x = d
// NR interleaved vectors (entry ((row NR + k) DIM + c), the batched PCG's layout): one lane per (row, vector) pair; NR = 1 is the plain kernel
template <int DIM, int NR = 1>
__global__ void __launch_bounds__(256) k_mg_cheb(int64_t nRows, const double *__restrict__ dinv, const double *__restrict__ rin,
                                                 const double *__restrict__ t, double *__restrict__ rout, double *__restrict__ d,
                                                 double *__restrict__ x, double a, double b, int first, int assign, MgGate g) {
    if (mg_closed(g)) return;
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nRows * NR; n += (int64_t)gridDim.x * 256) {
        double rv[DIM], zv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) rv[c] = rin[n * DIM + c] - (t ? t[n * DIM + c] : 0.0);
        if (rout)
#pragma unroll
            for (int c = 0; c < DIM; ++c) rout[n * DIM + c] = rv[c];
        apply_block<DIM>(dinv + (n / NR) * (DIM * (DIM + 1) / 2), rv, zv);
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const double dv = (first ? 0.0 : a * d[n * DIM + c]) + b * zv[c];
            if (d) d[n * DIM + c] = dv;                    // (a single-step sweep needs no direction vector)
            x[n * DIM + c] = assign ? dv : x[n * DIM + c] + dv;
        }
    }
}

// The V-cycle's LAST kernel with the PCG's inner product folded in (one-step smoother on the first level; mg_precond's MgFuse):
//   x += b D^-1 (rin - t) ;  x = rin on the fixed variables (k_mg_rz's rule) ;  scal[(it + 1) 4] += rin . x
// rin IS the PCG residual here (a one-step sweep smooths the right-hand side itself), so k_mg_rz's two vector reads disappear. A lane takes two
// consecutive rows in 16-byte accesses like k_pcg_update.
template <int DIM, class DinvT = double>
__global__ void __launch_bounds__(256) k_mg_cheb_rz(int64_t nRows, const DinvT *__restrict__ dinv, const double *__restrict__ rin,
                                                    const double *__restrict__ t, double *__restrict__ x, double b,
                                                    const uint8_t *__restrict__ mask, double *scalOut, MgGate g, DetBuf det) {
    __shared__ double red[8];
    if (mg_closed(g)) return;
    const int it = g.scal ? g.it + (int)g.stop[3] : g.it;
    constexpr int NS = DIM * (DIM + 1) / 2;
    double acc[1] = {0};
    const int64_t nPair = nRows >> 1;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nPair; q += (int64_t)gridDim.x * 256) {
        double rv[2 * DIM], tv[2 * DIM], xv[2 * DIM], zv[2 * DIM], dm[2 * NS];
        const double2 *r2 = reinterpret_cast<const double2 *>(rin + q * 2 * DIM);
        const double2 *t2 = reinterpret_cast<const double2 *>(t + q * 2 * DIM);
        double2 *x2 = reinterpret_cast<double2 *>(x + q * 2 * DIM);
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const double2 a = r2[c], u = t2[c], w = x2[c];
            rv[2 * c] = a.x; rv[2 * c + 1] = a.y; tv[2 * c] = a.x - u.x; tv[2 * c + 1] = a.y - u.y; xv[2 * c] = w.x; xv[2 * c + 1] = w.y;
        }
        load_dinv_pair<NS>(dinv + q * 2 * NS, dm);
        apply_block<DIM>(dm, tv, zv);
        apply_block<DIM>(dm + NS, tv + DIM, zv + DIM);
#pragma unroll
        for (int c = 0; c < 2 * DIM; ++c) {
            xv[c] += b * zv[c];
            if (mask && mask[q * 2 * DIM + c]) xv[c] = rv[c];
            acc[0] += rv[c] * xv[c];
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) x2[c] = make_double2(xv[2 * c], xv[2 * c + 1]);
    }
    if ((nRows & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // the odd last row
        const int64_t n = nRows - 1;
        double rv[DIM], tv[DIM], zv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) { rv[c] = rin[n * DIM + c]; tv[c] = rv[c] - t[n * DIM + c]; }
        apply_block_t<DIM, DinvT>(dinv + n * NS, tv, zv);
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            double xv = x[n * DIM + c] + b * zv[c];
            if (mask && mask[n * DIM + c]) xv = rv[c];
            x[n * DIM + c] = xv;
            acc[0] += rv[c] * xv;
        }
    }
    block_sum<1>(acc, red);
    double *const tg[1] = {scalOut + (int64_t)(it + 1) * 4};
    commit_sums<1>(acc, tg, det, red);
}

// restriction R = P^T of the residual r - t (t may be null): coarse DoF q gets its own fine DoF plus half of every
// edge-node DoF it is an end of. A CSR row of ~13 scattered entries per coarse DoF: EIGHT lanes share a row (every eighth entry
// each, independent loads in flight) and add up with three shuffles -- one lane per row walked its 13 dependent-latency loads alone
// and took 0.66 ms at config 3, more than an application of the quadratic operator.
#ifndef MG_RESTRICT_LANES
#define MG_RESTRICT_LANES 8     // lanes per coarse row: 4 / 8 / 16 -> 316 / 274 / 291 us at config 3 (rocprofv3, same box)
#endif
// NR right-hand sides: the FINE vectors are separate (vector k at r + k fineStride: the quadratic level runs one PCG loop per right-hand side),
// the COARSE ones interleaved (entry ((row NR + k) DIM + c): the linear level and everything below it serve all NR at once)
// (HAST: t != null, decided once per launch -- a test per component inside the gather loop kept the loads of an entry from being issued together.
// One right-hand side: TWO entries per lane and trip -- their index loads, then all their value loads, are in flight together.)
template <int DIM, int NR, bool HAST>
DEV void mg_restrict_body(int64_t nCoarse, const int32_t *__restrict__ fineOf, const int32_t *__restrict__ resPtr, const int32_t *__restrict__ resIdx,
                          const double *__restrict__ r, const double *__restrict__ t, int64_t fineStride, const uint8_t *__restrict__ coarseMask,
                          double *__restrict__ rc) {
    constexpr int W = NR * DIM;
    constexpr int L = MG_RESTRICT_LANES;
    const int sub = threadIdx.x & (L - 1);
    const int64_t nq = (nCoarse + 63) / 64 * 64;                 // whole waves take part in the shuffles
    for (int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) / L; q < nq; q += ((int64_t)gridDim.x * 256) / L) {
        double acc[W];
#pragma unroll
        for (int c = 0; c < W; ++c) acc[c] = 0.0;
        if (q < nCoarse) {
            const int k0 = resPtr[q], k1 = resPtr[q + 1];
            constexpr int U = NR == 1 ? 2 : 1;      // (two entries of NR = 6 vectors are 154 VGPRs: 900 against 670 us at configs[3])
            for (int k = k0 + sub; k < k1; k += U * L) {
                const bool two = U == 2 && k + L < k1;
                const int64_t e0 = resIdx[k], e1 = two ? resIdx[k + L] : e0;
                double v0[W], v1[U == 2 ? W : 1];
#pragma unroll
                for (int v = 0; v < NR; ++v)
#pragma unroll
                    for (int c = 0; c < DIM; ++c) {
                        const int64_t i0 = v * fineStride + e0 * DIM + c, i1 = v * fineStride + e1 * DIM + c;
                        v0[v * DIM + c] = HAST ? r[i0] - t[i0] : r[i0];
                        if (U == 2) v1[v * DIM + c] = HAST ? r[i1] - t[i1] : r[i1];
                    }
                const double w1 = two ? 0.5 : 0.0;
#pragma unroll
                for (int c = 0; c < W; ++c) acc[c] += U == 2 ? 0.5 * v0[c] + w1 * v1[c] : 0.5 * v0[c];
            }
            if (sub == 0) {
                const int64_t f = fineOf[q];
#pragma unroll
                for (int v = 0; v < NR; ++v)
#pragma unroll
                    for (int c = 0; c < DIM; ++c) {
                        const int64_t i = v * fineStride + f * DIM + c;
                        acc[v * DIM + c] += HAST ? r[i] - t[i] : r[i];
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < W; ++c) {
#pragma unroll
            for (int m = 1; m < L; m <<= 1) acc[c] += __shfl_xor(acc[c], m, 64);
        }
        if (q < nCoarse && sub == 0)
#pragma unroll
            for (int c = 0; c < W; ++c) rc[q * W + c] = (coarseMask && coarseMask[q * DIM + c % DIM]) ? 0.0 : acc[c];
    }
}
template <int DIM, int NR = 1>
__global__ void __launch_bounds__(256) k_mg_restrict(int64_t nCoarse, const int32_t *__restrict__ fineOf, const int32_t *__restrict__ resPtr,
                                                     const int32_t *__restrict__ resIdx, const double *__restrict__ r,
                                                     const double *__restrict__ t, int64_t fineStride, const uint8_t *__restrict__ coarseMask,
                                                     double *__restrict__ rc, MgGate g) {
    if (mg_closed(g)) return;
    if (t) mg_restrict_body<DIM, NR, true>(nCoarse, fineOf, resPtr, resIdx, r, t, fineStride, coarseMask, rc);
    else mg_restrict_body<DIM, NR, false>(nCoarse, fineOf, resPtr, resIdx, r, t, fineStride, coarseMask, rc);
}

// x += P xc on the free fine variables
// (ldc: doubles between consecutive coarse rows -- DIM, or NR DIM when xc points at vector k of NR interleaved coarse vectors)
template <int DIM>
__global__ void __launch_bounds__(256) k_mg_prolong_add(int64_t nFine, const int32_t *__restrict__ parA, const int32_t *__restrict__ parB,
                                                        const double *__restrict__ xc, int ldc, const uint8_t *__restrict__ fineMask,
                                                        double *__restrict__ x, MgGate g) {
    if (mg_closed(g)) return;
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nFine; n += (int64_t)gridDim.x * 256) {
        const int64_t a = parA[n], b = parB[n];
        if (a < 0) continue;
        double xa[DIM], xb[DIM], xv[DIM];       // all loads first, unconditionally (a test per component kept them apart); the mask gates the stores
#pragma unroll
        for (int c = 0; c < DIM; ++c) { xa[c] = xc[a * ldc + c]; xb[c] = xc[b * ldc + c]; xv[c] = x[n * DIM + c]; }
        uint8_t fx[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) fx[c] = fineMask ? fineMask[n * DIM + c] : (uint8_t)0;
#pragma unroll
        for (int c = 0; c < DIM; ++c)
            if (!fx[c]) x[n * DIM + c] = xv[c] + 0.5 * (xa[c] + xb[c]);
    }
}

// The same for the NR right-hand sides of the batched V-cycle in ONE launch: coarse vectors interleaved (entry ((row NR + k) DIM + c)), fine
// vectors separate (vector k at x + k vecStride); one lane per (fine row, vector) pair -- the NR lanes of a row read 24 NR contiguous bytes of each
// parent. Loop k's own gate (history scal + k scalStride, control block stop + 4 k) freezes its vector once it has converged.
template <int DIM, int NR>
__global__ void __launch_bounds__(256) k_mg_prolong_add_nr(int64_t nFine, const int32_t *__restrict__ parA, const int32_t *__restrict__ parB,
                                                           const double *__restrict__ xc, const uint8_t *__restrict__ fineMask, double *__restrict__ x,
                                                           int64_t vecStride, const double *scal, int64_t scalStride, int it, const double *stop) {
    __shared__ int sOpen[NR];
    if (threadIdx.x < NR) {
        int open = 1;
        if (scal) {
            const double *st = stop + 4 * threadIdx.x;
            const int itk = it + (int)st[3];
            open = !(itk >= 0 && scal[threadIdx.x * scalStride + (int64_t)itk * 4 + 2] <= st[0]);
        }
        sOpen[threadIdx.x] = open;
    }
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nFine * NR; p += (int64_t)gridDim.x * 256) {
        const int64_t n = p / NR;
        const int k = (int)(p - n * NR);
        if (!sOpen[k]) continue;
        const int64_t a = parA[n], b = parB[n];
        if (a < 0) continue;
        double xa[DIM], xb[DIM], xv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) { xa[c] = xc[(a * NR + k) * DIM + c]; xb[c] = xc[(b * NR + k) * DIM + c]; xv[c] = x[k * vecStride + n * DIM + c]; }
        uint8_t fx[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) fx[c] = fineMask ? fineMask[n * DIM + c] : (uint8_t)0;
#pragma unroll
        for (int c = 0; c < DIM; ++c)
            if (!fx[c]) x[k * vecStride + n * DIM + c] = xv[c] + 0.5 * (xa[c] + xb[c]);
    }
}

// x += Z yc on the free variables (prolongation of the rigid-body coarse correction, linear level)
// (coarse vectors of NR right-hand sides are [coarse index][NR], like the two-level kernels')
template <int DIM, int NR = 1>
__global__ void __launch_bounds__(256) k_mg_tl_prolong_add(TLArgs t, const double *__restrict__ yc, double *__restrict__ x, double alpha, MgGate g) {
    if (mg_closed(g)) return;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < t.nDoF * NR; p += (int64_t)gridDim.x * 256) {
        const int64_t n = p / NR;
        const int kr = (int)(p - n * NR);
        const int a = t.aggOfDof[n];
        const double rp[3] = {t.relPos[n * 3], t.relPos[n * 3 + 1], t.relPos[n * 3 + 2]};
        double zv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) zv[c] = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (k >= t.nModes) break;
            const double y = yc[((int64_t)a * t.nModes + k) * NR + kr];
#pragma unroll
            for (int c = 0; c < DIM; ++c) zv[c] += y * tl_mode<DIM>(k, c, rp);
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c)
            if (!(t.fixedMask && t.fixedMask[n * DIM + c])) x[p * DIM + c] += alpha * zv[c];
    }
}

// ------------------------------------------------------------------------------------------------
// Aggregate levels of the multigrid hierarchy: NM rigid-body modes per aggregate (6 in 3D, 3 in 2D), operators in LATTICE-STENCIL
// storage -- aggregates are the non-empty bins of a uniform lattice, an element spans at most adjacent bins, so row aggregate a
// couples to its NS = 3^dim lattice neighbours only: A[(a NS + slot) NM^2 + k NM + l], nbr[a NS + slot] = neighbour's id or -1.
// Coarser levels merge 2^dim bins; a parent mode (t, w) seen from a child reads  t_c = t + w x rho,  w_c = sc w  with
// rho = (c_child - c_parent) / H_parent and sc = H_child / H_parent (rotations are scaled by 1 / H like tl_mode).
// ------------------------------------------------------------------------------------------------
template <int DIM> struct StDims { static constexpr int NM = DIM == 3 ? 6 : 3, NS = DIM == 3 ? 27 : 9; };

// child = T parent (NM vectors): rel = {rho_x, rho_y, rho_z, sc}
template <int DIM> DEV void st_transfer(const double *rel, const double *par, double *child) {
    if (DIM == 3) {
        const double rx = rel[0], ry = rel[1], rz = rel[2], sc = rel[3];
        const double wx = par[3], wy = par[4], wz = par[5];
        child[0] = par[0] + (wy * rz - wz * ry);
        child[1] = par[1] + (wz * rx - wx * rz);
        child[2] = par[2] + (wx * ry - wy * rx);
        child[3] = sc * wx; child[4] = sc * wy; child[5] = sc * wz;
    } else {
        const double rx = rel[0], ry = rel[1], sc = rel[3], w = par[2];      // u = t + w (-r_y, r_x)
        child[0] = par[0] - w * ry;
        child[1] = par[1] + w * rx;
        child[2] = sc * w;
    }
}
// parent += T^T child
template <int DIM> DEV void st_transfer_T(const double *rel, const double *ch, double *par) {
    if (DIM == 3) {
        const double rx = rel[0], ry = rel[1], rz = rel[2], sc = rel[3];
        par[0] += ch[0]; par[1] += ch[1]; par[2] += ch[2];
        // (w x rho) . f = w . (rho x f)
        par[3] += sc * ch[3] + (ry * ch[2] - rz * ch[1]);
        par[4] += sc * ch[4] + (rz * ch[0] - rx * ch[2]);
        par[5] += sc * ch[5] + (rx * ch[1] - ry * ch[0]);
    } else {
        const double rx = rel[0], ry = rel[1], sc = rel[3];
        par[0] += ch[0]; par[1] += ch[1];
        par[2] += sc * ch[2] + (rx * ch[1] - ry * ch[0]);
    }
}

// NR right-hand sides: vectors are [aggregate][mode][NR] (the layout of the two-level kernels' coarse vectors); one lane per (aggregate, mode,
// vector) -- the NR lanes of a matrix row read the same stencil entries (one pass over the matrix for all NR vectors)
template <int DIM, int NR = 1>
__global__ void __launch_bounds__(256) k_st_spmv(int64_t nAgg, const int32_t *__restrict__ nbr, const double *__restrict__ A, const float *__restrict__ A32,
                                                 const double *__restrict__ x, double *__restrict__ y, MgGate g) {
    constexpr int NM = StDims<DIM>::NM, NS = StDims<DIM>::NS;
    if (mg_closed(g)) return;
    for (int64_t q0 = (int64_t)blockIdx.x * 256 + threadIdx.x; q0 < nAgg * NM * NR; q0 += (int64_t)gridDim.x * 256) {
        const int64_t q = q0 / NR;
        const int kr = (int)(q0 - q * NR);
        const int64_t a = q / NM;
        const int k = (int)(q - a * NM);
        double acc = 0;
        if (A32) {                       // the stencil blocks rounded to FP32 (option mg_coarse_fp32): half the bytes this kernel is bound by
            for (int sl = 0; sl < NS; ++sl) {
                const int b = nbr[a * NS + sl];
                if (b < 0) continue;
                const float *row = A32 + ((a * NS + sl) * NM + k) * NM;
#pragma unroll
                for (int l = 0; l < NM; ++l) acc += (double)row[l] * x[((int64_t)b * NM + l) * NR + kr];
            }
        } else
        for (int sl = 0; sl < NS; ++sl) {
            const int b = nbr[a * NS + sl];
            if (b < 0) continue;
            const double *row = A + ((a * NS + sl) * NM + k) * NM;
#pragma unroll
            for (int l = 0; l < NM; ++l) acc += row[l] * x[((int64_t)b * NM + l) * NR + kr];
        }
        y[q0] = acc;
    }
}

// inverse of the diagonal NM x NM blocks; a mode without stiffness of its own (empty direction, all its DoFs fixed, a one-node
// aggregate's rotations) is dropped: its row and column of the inverse are zero, the level leaves it to the others
template <int DIM>
__global__ void __launch_bounds__(256) k_st_dinv(int64_t nAgg, const double *__restrict__ A, double *__restrict__ Dinv) {
    constexpr int NM = StDims<DIM>::NM, NS = StDims<DIM>::NS;
    for (int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x; a < nAgg; a += (int64_t)gridDim.x * 256) {
        double M[NM][2 * NM];
        const double *D = A + (a * NS + NS / 2) * NM * NM;
        double dmax = 0;
#pragma unroll
        for (int i = 0; i < NM; ++i) dmax = fmax(dmax, D[i * NM + i]);
        bool dead[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) dead[i] = !(D[i * NM + i] > 1e-10 * dmax);
#pragma unroll
        for (int i = 0; i < NM; ++i)
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                M[i][j] = (dead[i] || dead[j]) ? (i == j ? 1.0 : 0.0) : 0.5 * (D[i * NM + j] + D[j * NM + i]);
                M[i][NM + j] = i == j ? 1.0 : 0.0;
            }
        // Gauss-Jordan without pivoting (SPD); a pivot that collapses marks a dependent mode, dropped like a dead one
        for (int c = 0; c < NM; ++c) {
            double p = M[c][c];
            if (!(p > 1e-12 * dmax) && !dead[c]) {
                dead[c] = true;
                for (int j = 0; j < 2 * NM; ++j) M[c][j] = 0.0;
                for (int i = 0; i < NM; ++i) M[i][c] = 0.0;
                M[c][c] = 1.0; M[c][NM + c] = 1.0;
                p = 1.0;
            }
            const double inv = 1.0 / p;
            for (int j = 0; j < 2 * NM; ++j) M[c][j] *= inv;
            for (int i = 0; i < NM; ++i) {
                if (i == c) continue;
                const double f = M[i][c];
                if (f != 0.0) for (int j = 0; j < 2 * NM; ++j) M[i][j] -= f * M[c][j];
            }
        }
#pragma unroll
        for (int i = 0; i < NM; ++i)
#pragma unroll
            for (int j = 0; j < NM; ++j) Dinv[a * NM * NM + i * NM + j] = (dead[i] || dead[j]) ? 0.0 : M[i][NM + j];
    }
}

// Chebyshev step on an aggregate level (see k_mg_cheb): block size NM, full inverse blocks
template <int DIM, int NR = 1>
__global__ void __launch_bounds__(256) k_st_cheb(int64_t nAgg, const double *__restrict__ Dinv, const double *__restrict__ rin, const double *__restrict__ t,
                                                 double *__restrict__ rout, double *__restrict__ d, double *__restrict__ x, double ca, double cb, int first,
                                                 int assign, MgGate g) {
    constexpr int NM = StDims<DIM>::NM;
    if (mg_closed(g)) return;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nAgg * NR; p += (int64_t)gridDim.x * 256) {
        const int64_t a = p / NR;
        const int kr = (int)(p - a * NR);
        double rv[NM];
#pragma unroll
        for (int k = 0; k < NM; ++k) rv[k] = rin[(a * NM + k) * NR + kr] - (t ? t[(a * NM + k) * NR + kr] : 0.0);
        if (rout)
#pragma unroll
            for (int k = 0; k < NM; ++k) rout[(a * NM + k) * NR + kr] = rv[k];
#pragma unroll
        for (int k = 0; k < NM; ++k) {
            double z = 0;
#pragma unroll
            for (int l = 0; l < NM; ++l) z += Dinv[a * NM * NM + k * NM + l] * rv[l];
            const int64_t o = (a * NM + k) * NR + kr;
            const double dv = (first ? 0.0 : ca * d[o]) + cb * z;
            if (d) d[o] = dv;
            x[o] = assign ? dv : x[o] + dv;
        }
    }
}

// Galerkin product between two aggregate levels: Ac[p][offset] = sum over the children a of p and their stencil slots towards a child b of
// the parent at that lattice offset of T_a^T A[a][slot] T_b. GATHER form: one lane per (parent, offset) walks the parent's children in
// index order and keeps the 6 x 6 sum in registers -- no atomics, no zero fill, the same bits on every run (the scatter form this replaces
// added with unsafeAtomicAdd in arrival order).
template <int DIM>
__global__ void __launch_bounds__(256) k_st_rap(int64_t nParents, const int32_t *__restrict__ childPtr, const int32_t *__restrict__ childIdx,
                                                const int32_t *__restrict__ nbr, const double *__restrict__ A, const int32_t *__restrict__ parent,
                                                const double *__restrict__ rel, const int32_t *__restrict__ coordC, double *__restrict__ Ac, int wrapX, int wrapY,
                                                int wrapZ) {
    constexpr int NM = StDims<DIM>::NM, NS = StDims<DIM>::NS;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nParents * NS; q += (int64_t)gridDim.x * 256) {
        const int64_t pa = q / NS;
        const int cs = (int)(q - pa * NS);
        double C[NM][NM];
#pragma unroll
        for (int k = 0; k < NM; ++k)
#pragma unroll
            for (int l = 0; l < NM; ++l) C[k][l] = 0.0;
        for (int ci = childPtr[pa]; ci < childPtr[pa + 1]; ++ci) {
            const int64_t a = childIdx[ci];
            for (int slot = 0; slot < NS; ++slot) {
                const int b = nbr[a * NS + slot];
                if (b < 0) continue;
                const int pb = parent[b];
                const int dx = lattice_wrap(coordC[pb * 3] - coordC[pa * 3], wrapX), dy = lattice_wrap(coordC[pb * 3 + 1] - coordC[pa * 3 + 1], wrapY),
                          dz = lattice_wrap(coordC[pb * 3 + 2] - coordC[pa * 3 + 2], wrapZ);
                if ((dx + 1) + 3 * (dy + 1) + (DIM == 3 ? 9 * (dz + 1) : 0) != cs) continue;
                const double *B = A + (a * NS + slot) * NM * NM;
                // W = B T_b (columns: parent modes of b), then C += T_a^T W
                double W[NM][NM];
#pragma unroll
                for (int l = 0; l < NM; ++l) {
                    double e[NM], tc[NM];
#pragma unroll
                    for (int m = 0; m < NM; ++m) e[m] = m == l ? 1.0 : 0.0;
                    st_transfer<DIM>(rel + (int64_t)b * 4, e, tc);
#pragma unroll
                    for (int k = 0; k < NM; ++k) {
                        double v = 0;
#pragma unroll
                        for (int m = 0; m < NM; ++m) v += B[k * NM + m] * tc[m];
                        W[k][l] = v;
                    }
                }
#pragma unroll
                for (int l = 0; l < NM; ++l) {
                    double col[NM], out[NM];
#pragma unroll
                    for (int k = 0; k < NM; ++k) { col[k] = W[k][l]; out[k] = 0.0; }
                    st_transfer_T<DIM>(rel + a * 4, col, out);
#pragma unroll
                    for (int k = 0; k < NM; ++k) C[k][l] += out[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NM; ++k)
#pragma unroll
            for (int l = 0; l < NM; ++l) Ac[(q * NM + k) * NM + l] = C[k][l];
    }
}

// restriction between two aggregate levels, gather form: rc[p] = sum over the children a of p, in index order, of T_a^T (r - t)[a]
template <int DIM, int NR = 1>
__global__ void __launch_bounds__(256) k_st_restrict(int64_t nParents, const int32_t *__restrict__ childPtr, const int32_t *__restrict__ childIdx,
                                                     const double *__restrict__ rel, const double *__restrict__ r, const double *__restrict__ t,
                                                     double *__restrict__ rc, MgGate g) {
    constexpr int NM = StDims<DIM>::NM;
    if (mg_closed(g)) return;
    for (int64_t p0 = (int64_t)blockIdx.x * 256 + threadIdx.x; p0 < nParents * NR; p0 += (int64_t)gridDim.x * 256) {
        const int64_t p = p0 / NR;
        const int kr = (int)(p0 - p * NR);
        double out[NM];
#pragma unroll
        for (int k = 0; k < NM; ++k) out[k] = 0.0;
        for (int ci = childPtr[p]; ci < childPtr[p + 1]; ++ci) {
            const int64_t a = childIdx[ci];
            double ch[NM];
#pragma unroll
            for (int k = 0; k < NM; ++k) ch[k] = r[(a * NM + k) * NR + kr] - (t ? t[(a * NM + k) * NR + kr] : 0.0);
            st_transfer_T<DIM>(rel + a * 4, ch, out);
        }
#pragma unroll
        for (int k = 0; k < NM; ++k) rc[(p * NM + k) * NR + kr] = out[k];
    }
}

template <int DIM, int NR = 1>
__global__ void __launch_bounds__(256) k_st_prolong_add(int64_t nAgg, const int32_t *__restrict__ parent, const double *__restrict__ rel, const double *__restrict__ xc,
                                                        double *__restrict__ x, double alpha, MgGate g) {
    constexpr int NM = StDims<DIM>::NM;
    if (mg_closed(g)) return;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < nAgg * NR; p += (int64_t)gridDim.x * 256) {
        const int64_t a = p / NR;
        const int kr = (int)(p - a * NR);
        double par[NM], ch[NM];
        const int64_t pa = parent[a];
#pragma unroll
        for (int k = 0; k < NM; ++k) par[k] = xc[(pa * NM + k) * NR + kr];
        st_transfer<DIM>(rel + a * 4, par, ch);
#pragma unroll
        for (int k = 0; k < NM; ++k) x[(a * NM + k) * NR + kr] += alpha * ch[k];
    }
}

// stencil -> dense row-major m x m (the coarsest level is inverted densely)
template <int DIM>
__global__ void __launch_bounds__(256) k_st_to_dense(int64_t nAgg, const int32_t *__restrict__ nbr, const double *__restrict__ A, double *__restrict__ Ad) {
    constexpr int NM = StDims<DIM>::NM, NS = StDims<DIM>::NS;
    const int64_t m = nAgg * NM;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nAgg * NS * NM * NM; q += (int64_t)gridDim.x * 256) {
        const int64_t as = q / (NM * NM);
        const int e = (int)(q - as * NM * NM);
        const int b = nbr[as];
        if (b < 0) continue;
        const int64_t a = as / NS;
        Ad[(a * NM + e / NM) * m + (int64_t)b * NM + e % NM] = A[q];
    }
}

// gated zero fill
__global__ void __launch_bounds__(256) k_mg_zero(int64_t n, double *__restrict__ v, MgGate g) {
    if (mg_closed(g)) return;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) v[k] = 0.0;
}

// pseudo-random values in (-1/2, 1/2) (start vector of the power iteration)
__global__ void __launch_bounds__(256) k_fill_hash(int64_t n, double *__restrict__ v) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        uint64_t z = (uint64_t)k * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull;     // splitmix64 finaliser
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z ^= z >> 31;
        v[k] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}

// dst = (float)src
__global__ void __launch_bounds__(256) k_to_f32(int64_t n, const double *__restrict__ src, float *__restrict__ dst) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) dst[k] = (float)src[k];
}
// dst[e][0..w) = src[e][0..w): the corner columns of a node table (linear level of the multigrid hierarchy on the device)
__global__ void __launch_bounds__(256) k_take_columns_i32(int64_t n, int W, int w, const int32_t *__restrict__ src, int32_t *__restrict__ dst) {
    const int64_t total = n * w;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t e = k / w;
        dst[k] = src[e * W + (k - e * w)];
    }
}

// out = a - b (b may be null: copy), gated
__global__ void __launch_bounds__(256) k_mg_diff(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out, MgGate g) {
    if (mg_closed(g)) return;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) out[k] = a[k] - (b ? b[k] : 0.0);
}

// scal[(it + 1) 4] += r . z  (the preconditioned inner product of the PCG), z = r on the fixed variables first
__global__ void __launch_bounds__(256) k_mg_rz(int64_t n, const double *__restrict__ r, double *__restrict__ z, const uint8_t *__restrict__ mask,
                                               double *scal, MgGate g, DetBuf det) {
    __shared__ double red[8];
    if (mg_closed(g)) return;
    const int it = g.scal ? g.it + (int)g.stop[3] : g.it;
    double acc[1] = {0};
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) {
        double zv = z[k];
        if (mask && mask[k]) { zv = r[k]; z[k] = zv; }
        acc[0] += r[k] * zv;
    }
    block_sum<1>(acc, red);
    double *const tg[1] = {scal + (int64_t)(it + 1) * 4};
    commit_sums<1>(acc, tg, det, red);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void launch_untile_vals(int dim, int64_t nnzb, const double *tiled, double *aos, hipStream_t s) {
    if (!nnzb) return;
    hipLaunchKernelGGL(k_untile, dim3(grid_for(nnzb * dim * dim, 8192)), dim3(256), 0, s, dim * dim, nnzb, tiled, aos);
    CHECK_LAUNCH();
}

void launch_extract_diag_inv(int dim, int64_t nRows, const int32_t *rowPtr, const int32_t *colIdx, const double *vals,
                             const uint8_t *fixedMask, int kind, double *dinv, hipStream_t s) {
    const int grid = (int)((nRows + 255) / 256);
    if (dim == 1) hipLaunchKernelGGL(k_diag_inv<1>, dim3(grid), dim3(256), 0, s, nRows, rowPtr, colIdx, vals, fixedMask, kind, dinv);
    else if (dim == 3) hipLaunchKernelGGL(k_diag_inv<3>, dim3(grid), dim3(256), 0, s, nRows, rowPtr, colIdx, vals, fixedMask, kind, dinv);
    else hipLaunchKernelGGL(k_diag_inv<2>, dim3(grid), dim3(256), 0, s, nRows, rowPtr, colIdx, vals, fixedMask, kind, dinv);
    CHECK_LAUNCH();
}

void launch_precond(int dim, int64_t nRows, const double *dinv, const double *r, double *z, hipStream_t s) {
    if (dim == 1) hipLaunchKernelGGL(k_precond<1>, dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, r, z);
    else if (dim == 3) hipLaunchKernelGGL(k_precond<3>, dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, r, z);
    else hipLaunchKernelGGL(k_precond<2>, dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, r, z);
    CHECK_LAUNCH();
}

void launch_pcg_init(int dim, int64_t nRows, const double *dinv, const double *b, double *x, double *r, double *z, double *p,
                     double *scal, hipStream_t s) {
    if (dim == 1) hipLaunchKernelGGL(k_pcg_init<1>, dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, b, x, r, z, p, scal, t_det);
    else if (dim == 3) hipLaunchKernelGGL(k_pcg_init<3>, dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, b, x, r, z, p, scal, t_det);
    else hipLaunchKernelGGL(k_pcg_init<2>, dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, b, x, r, z, p, scal, t_det);
    launch_det_finish(s);
    CHECK_LAUNCH();
}

void launch_pcg_update(int dim, int64_t nRows, const double *dinv, const double *Ap, double *r,
                       double *z, double *scal, int it, const double *stopPtr, hipStream_t s) {
    const int grid = grid_for(nRows / 2);   // 2048 workgroups at most: each ends with two atomics on the same two scalars
    if (dim == 1) hipLaunchKernelGGL((k_pcg_update<1, false>), dim3(grid), dim3(256), 0, s, nRows, dinv, Ap, r, z, scal, it, stopPtr, t_det);
    else if (dim == 3) hipLaunchKernelGGL((k_pcg_update<3, false>), dim3(grid), dim3(256), 0, s, nRows, dinv, Ap, r, z, scal, it, stopPtr, t_det);
    else hipLaunchKernelGGL((k_pcg_update<2, false>), dim3(grid), dim3(256), 0, s, nRows, dinv, Ap, r, z, scal, it, stopPtr, t_det);
    launch_det_finish(s);
    CHECK_LAUNCH();
}

void launch_pcg_update_noz(int dim, int64_t nRows, const double *Ap, double *r, double *scal, int it,
                           const double *stopPtr, hipStream_t s) {
    const int grid = grid_for(nRows / 2);
    if (dim == 3) hipLaunchKernelGGL((k_pcg_update<3, true>), dim3(grid), dim3(256), 0, s, nRows, (const double *)nullptr, Ap, r, (double *)nullptr, scal, it, stopPtr, t_det);
    else hipLaunchKernelGGL((k_pcg_update<2, true>), dim3(grid), dim3(256), 0, s, nRows, (const double *)nullptr, Ap, r, (double *)nullptr, scal, it, stopPtr, t_det);
    launch_det_finish(s);
    CHECK_LAUNCH();
}

// r -= alpha Ap, z = zs Dinv r, r.r (no r.z): the PCG update with the V-cycle's pre-smoothing from zero folded in
void launch_pcg_update_presmooth(int dim, int64_t nRows, const double *dinv, const float *dinv32, const double *Ap, double *r, double *z, double zs, double *scal, int it,
                                 const double *stopPtr, hipStream_t s) {
    const int grid = grid_for(nRows / 2);
    if (dinv32) {           // the smoother's FP32 copy of the inverse diagonal blocks: 5 vector passes instead of 6
        if (dim == 3) hipLaunchKernelGGL((k_pcg_update<3, false, true, float>), dim3(grid), dim3(256), 0, s, nRows, dinv32, Ap, r, z, scal, it, stopPtr, t_det, zs);
        else if (dim == 2) hipLaunchKernelGGL((k_pcg_update<2, false, true, float>), dim3(grid), dim3(256), 0, s, nRows, dinv32, Ap, r, z, scal, it, stopPtr, t_det, zs);
        else hipLaunchKernelGGL((k_pcg_update<1, false, true, float>), dim3(grid), dim3(256), 0, s, nRows, dinv32, Ap, r, z, scal, it, stopPtr, t_det, zs);
    } else
    if (dim == 3) hipLaunchKernelGGL((k_pcg_update<3, false, true>), dim3(grid), dim3(256), 0, s, nRows, dinv, Ap, r, z, scal, it, stopPtr, t_det, zs);
    else if (dim == 2) hipLaunchKernelGGL((k_pcg_update<2, false, true>), dim3(grid), dim3(256), 0, s, nRows, dinv, Ap, r, z, scal, it, stopPtr, t_det, zs);
    else hipLaunchKernelGGL((k_pcg_update<1, false, true>), dim3(grid), dim3(256), 0, s, nRows, dinv, Ap, r, z, scal, it, stopPtr, t_det, zs);
    launch_det_finish(s);
    CHECK_LAUNCH();
}

void launch_tl_fill(const TLArgs &t, const int32_t *colorOfAgg, int color, int mode, double *v, hipStream_t s) {
    if (t.dim == 3) hipLaunchKernelGGL(k_tl_fill<3>, dim3(grid_for(t.nDoF)), dim3(256), 0, s, t, colorOfAgg, color, mode, v);
    else hipLaunchKernelGGL(k_tl_fill<2>, dim3(grid_for(t.nDoF)), dim3(256), 0, s, t, colorOfAgg, color, mode, v);
    CHECK_LAUNCH();
}
void launch_tl_restrict(const TLArgs &t, const int32_t *aggPtr, const int32_t *dofsByAgg, const double *w, double *rc, hipStream_t s) {
    if (t.nDoF < (int64_t)t.nAgg * 160) {           // small aggregates: a wave each
        const unsigned grid = (unsigned)((t.nAgg + 3) / 4);
        if (t.dim == 3) hipLaunchKernelGGL(k_tl_restrict_wave<3>, dim3(grid), dim3(256), 0, s, t, aggPtr, dofsByAgg, w, rc);
        else hipLaunchKernelGGL(k_tl_restrict_wave<2>, dim3(grid), dim3(256), 0, s, t, aggPtr, dofsByAgg, w, rc);
    } else if (t.dim == 3) hipLaunchKernelGGL(k_tl_restrict<3>, dim3(t.nAgg), dim3(256), 0, s, t, aggPtr, dofsByAgg, w, rc);
    else hipLaunchKernelGGL(k_tl_restrict<2>, dim3(t.nAgg), dim3(256), 0, s, t, aggPtr, dofsByAgg, w, rc);
    CHECK_LAUNCH();
}
void launch_tl_scatter(int nAgg, int nModes, int nColor, const int32_t *nbrOfColor, int color, int mode, const double *R, double *Ac,
                       hipStream_t s) {
    hipLaunchKernelGGL(k_tl_scatter, dim3(grid_for((int64_t)nAgg * nModes)), dim3(256), 0, s, nAgg, nModes, nColor, nbrOfColor, color, mode, R, Ac);
    CHECK_LAUNCH();
}
void launch_tl_rap(const TLArgs &t, int64_t nRows, const int32_t *rowPtr, const int32_t *colIdx, const double *vals, double *Ac,
                   hipStream_t s) {
    const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((nRows + 3) / 4, 256 * 16));
    if (t.dim == 3) hipLaunchKernelGGL(k_tl_rap<3>, dim3(grid), dim3(256), 0, s, t, nRows, rowPtr, colIdx, vals, Ac);
    else hipLaunchKernelGGL(k_tl_rap<2>, dim3(grid), dim3(256), 0, s, t, nRows, rowPtr, colIdx, vals, Ac);
    CHECK_LAUNCH();
}
// Ac[i][j] and Ac[j][i] for entries of two DIFFERENT aggregates: both become the sum of the two partial values (see k_tl_rap_agg)
__global__ void __launch_bounds__(256) k_tl_mirror_upper(double *__restrict__ Ac, int64_t m, int NM) {
    const int64_t total = m * m;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t i = q / m, j = q - i * m;
        if (j <= i || i / NM == j / NM) continue;
        const double v = Ac[i * m + j] + Ac[j * m + i];
        Ac[i * m + j] = v;
        Ac[j * m + i] = v;
    }
}

// the same for the lattice-stencil storage: S[a][slot(b)] and S[b][slot(a)]^T (slot(a) seen from b is the opposite offset) both become the
// sum of the two partial sums; the pair is handled by the thread of its smaller aggregate
__global__ void __launch_bounds__(256) k_st_mirror_upper(double *__restrict__ S, const int32_t *__restrict__ nbr, int64_t nAgg, int NSLOT, int NM) {
    const int64_t total = nAgg * NSLOT * NM * NM;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < total; q += (int64_t)gridDim.x * 256) {
        const int64_t a = q / (NSLOT * NM * NM);
        const int r = (int)(q - a * NSLOT * NM * NM), sl = r / (NM * NM), e = r - sl * NM * NM, kk = e / NM, ll = e - kk * NM;
        if (sl == NSLOT / 2) continue;
        const int64_t b = nbr[a * NSLOT + sl];
        if (b < 0 || b <= a) continue;
        const int64_t ia = q, ib = (b * NSLOT + (NSLOT - 1 - sl)) * NM * NM + ll * NM + kk;
        const double v = S[ia] + S[ib];
        S[ia] = v;
        S[ib] = v;
    }
}
void launch_st_mirror_upper(double *stencil, const int32_t *nbr, int64_t nAgg, int dim, hipStream_t s) {
    const int NS = dim == 3 ? 27 : 9, NM = dim == 3 ? 6 : 3;
    hipLaunchKernelGGL(k_st_mirror_upper, dim3(grid_for(nAgg * NS * NM * NM)), dim3(256), 0, s, stencil, nbr, nAgg, NS, NM);
    CHECK_LAUNCH();
}

void launch_tl_rap_agg(const TLArgs &t, const int32_t *aggPtr, const int32_t *dofsByAgg, const int32_t *binCoord, const int32_t *rowPtr,
                       const int32_t *colIdx, const double *vals, double *Ac, hipStream_t s, bool upperOnly, int64_t nOwnedRows, double *stencil,
                       int *farCount, const int *wrapNb) {
    const int wx = wrapNb ? wrapNb[0] : 0, wy = wrapNb ? wrapNb[1] : 0, wz = wrapNb ? wrapNb[2] : 0;
    const int det = t_det.partials ? 1 : 0;
    const int NMl = t.dim == 3 ? 6 : 3, NSl = t.dim == 3 ? 27 : 9;
    const int threads = det ? 256 : 512;                    // deterministic: four waves, four neighbour tables (35 KB of LDS)
    const size_t lds = ((size_t)(det ? threads / 64 : 1) * NSl * NMl * NMl + (size_t)16 * NMl * NMl) * sizeof(double);
    if (t.dim == 3) hipLaunchKernelGGL(k_tl_rap_agg<3>, dim3(t.nAgg), dim3(threads), lds, s, t, aggPtr, dofsByAgg, binCoord, rowPtr, colIdx, vals, Ac, upperOnly ? 1 : 0, nOwnedRows, stencil, farCount, wx, wy, wz, det);
    else hipLaunchKernelGGL(k_tl_rap_agg<2>, dim3(t.nAgg), dim3(threads), lds, s, t, aggPtr, dofsByAgg, binCoord, rowPtr, colIdx, vals, Ac, upperOnly ? 1 : 0, nOwnedRows, stencil, farCount, wx, wy, wz, det);
    if (stencil) { CHECK_LAUNCH(); return; }
    if (upperOnly) {
        const int64_t m = (int64_t)t.nAgg * (t.dim == 3 ? 6 : 3);
        hipLaunchKernelGGL(k_tl_mirror_upper, dim3(grid_for(m * m)), dim3(256), 0, s, Ac, m, t.dim == 3 ? 6 : 3);
    }
    CHECK_LAUNCH();
}
// In-place-style dense SPD inverse: A (mp x mp, mp % 64 == 0) is overwritten by its Cholesky factor, the inverse
// goes to Ainv; X and Dinv are scratch (mp x mp and (mp/64) x 64 x 64). Returns false if A is not SPD.
bool dense_spd_inverse_device(double *A, double *X, double *Ainv, double *Dinv, int64_t mp, int *notSpdDev, hipStream_t s) {
    const int nt = (int)(mp / DT);
    MFH_HIP(hipMemsetAsync(notSpdDev, 0, sizeof(int), s));
    MFH_HIP(hipMemsetAsync(X, 0, sizeof(double) * mp * mp, s));
    for (int k = 0; k < nt; ++k) {
        hipLaunchKernelGGL(k_dense_potrf, dim3(1), dim3(256), 0, s, A, mp, k, Dinv, notSpdDev);
        const int rem = nt - k - 1;
        if (rem > 0) {
            hipLaunchKernelGGL(k_dense_trsm, dim3(rem), dim3(256), 0, s, A, mp, k, (const double *)Dinv);
            hipLaunchKernelGGL(k_dense_syrk, dim3(rem, rem), dim3(256), 0, s, A, mp, k);
        }
    }
    // L^-1: diagonal tiles from the factorisation, then recursive doubling (Ainv doubles as scratch until the last kernel)
    hipLaunchKernelGGL(k_dense_trinv, dim3(nt), dim3(256), 0, s, (const double *)A, X, mp, 0, (const double *)Dinv);
    for (int bt = 1; bt < nt; bt *= 2) {
        const dim3 grid(bt, bt, (nt + 2 * bt - 1) / (2 * bt));
        hipLaunchKernelGGL(k_dense_linv_level, grid, dim3(256), 0, s, (const double *)A, X, Ainv, mp, nt, bt, 1);
        hipLaunchKernelGGL(k_dense_linv_level, grid, dim3(256), 0, s, (const double *)A, X, Ainv, mp, nt, bt, 2);
    }
    hipLaunchKernelGGL(k_dense_xtx, dim3(nt, nt), dim3(256), 0, s, (const double *)X, Ainv, mp, nt);
    CHECK_LAUNCH();
    int bad = 0;
    MFH_HIP(hipMemcpyAsync(&bad, notSpdDev, sizeof(int), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    return bad == 0;
}

void launch_tl_prep(int64_t m, int64_t mp, const double *Ac, const uint8_t *dead, double maxd, double *Ap, hipStream_t s) {
    hipLaunchKernelGGL(k_tl_prep, dim3(grid_for(mp * mp, 16384)), dim3(256), 0, s, m, mp, Ac, dead, maxd, Ap);
    CHECK_LAUNCH();
}
void launch_tl_gemv(int64_t m, int64_t ld, const double *A, const double *x, double *y, hipStream_t s) {
    hipLaunchKernelGGL(k_tl_gemv, dim3((unsigned)m), dim3(256), 0, s, m, ld, A, x, y);
    CHECK_LAUNCH();
}
void launch_tl_apply(const TLArgs &t, const double *dinv, const double *r, const double *yc, double *z, double *scal, int it,
                     const double *stopPtr, hipStream_t s) {
    if (t.dim == 3) hipLaunchKernelGGL(k_tl_apply<3>, dim3(grid_for(t.nDoF)), dim3(256), 0, s, t, dinv, r, yc, z, scal, it, stopPtr, t_det);
    else hipLaunchKernelGGL(k_tl_apply<2>, dim3(grid_for(t.nDoF)), dim3(256), 0, s, t, dinv, r, yc, z, scal, it, stopPtr, t_det);
    launch_det_finish(s);
    CHECK_LAUNCH();
}

void launch_dev_update_xr(int64_t n, const double *num, const double *den, const double *p, const double *Ap, double *x, double *r, hipStream_t s) {
    hipLaunchKernelGGL(k_dev_update_xr, dim3(grid_for(n)), dim3(256), 0, s, n, num, den, p, Ap, x, r);
    CHECK_LAUNCH();
}
void launch_dev_direction(int64_t n, const double *num, const double *den, const double *z, double *p, hipStream_t s) {
    hipLaunchKernelGGL(k_dev_direction, dim3(grid_for(n)), dim3(256), 0, s, n, num, den, z, p);
    CHECK_LAUNCH();
}
void launch_dev_dots(int64_t n, const double *r, const double *z, double *out, hipStream_t s) {
    MFH_HIP(hipMemsetAsync(out, 0, 2 * sizeof(double), s));
    hipLaunchKernelGGL(k_dev_dots, dim3(grid_for(n)), dim3(256), 0, s, n, r, z, out, t_det);
    launch_det_finish(s);
    CHECK_LAUNCH();
}

void launch_add_scalar(double *p, double v, hipStream_t s) {
    hipLaunchKernelGGL(k_add_scalar, dim3(1), dim3(1), 0, s, p, v);
    CHECK_LAUNCH();
}
void launch_advance_base(double *stop, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_advance_base, dim3(1), dim3(1), 0, s, stop, (double)n);
    CHECK_LAUNCH();
}

void launch_pcg_direction(int64_t n, const double *z, double *p, double *x, const double *scal, int it, const double *stopPtr, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_direction, dim3(grid_for(n / 2, g_vecGridCap)), dim3(256), 0, s, n, z, p, x, scal, it, stopPtr);
    CHECK_LAUNCH();
}

void launch_axpby(int64_t n, double a, const double *x, double b, double *y, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_axpby, dim3(grid_for(n)), dim3(256), 0, s, n, a, x, b, y);
    CHECK_LAUNCH();
}
void launch_mask(int64_t n, const uint8_t *mask, double *v, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_mask, dim3(grid_for(n)), dim3(256), 0, s, n, mask, v);
    CHECK_LAUNCH();
}
void launch_scatter_values(int64_t n, const int64_t *idx, const double *val, double *v, int64_t bound, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_scatter_values, dim3(grid_for(n)), dim3(256), 0, s, n, idx, val, v, bound);
    CHECK_LAUNCH();
}
void launch_dot(int64_t n, const double *a, const double *b, double *out, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_dot, dim3(grid_for(n)), dim3(256), 0, s, n, a, b, out, t_det);
    launch_det_finish(s);
    CHECK_LAUNCH();
}



// ---- Chronopoulos-Gear PCG and its helpers
// grids of the pair-per-lane kernels: a multiple of 3 workgroups (768 lanes: a multiple of every supported NR)
static int pair_grid(int64_t nPairs) { return 3 * (int)std::max<int64_t>(1, std::min<int64_t>((nPairs + 767) / 768, 683)); }

void launch_cg_update(int dim, int64_t nRows, int NR, const double *dinv, double *u, const double *w, double *p, double *sv, double *x, double *r,
                      double *scal, int it, const double *ctl, bool skipU, hipStream_t s) {
    if (NR == 1) {
        const int g1 = grid_for(nRows / 2);
#define CALL1(D)                                                                                                                      \
    if (skipU) hipLaunchKernelGGL((k_cg_update1<D, true>), dim3(g1), dim3(256), 0, s, nRows, dinv, u, w, p, sv, x, r, scal, it, ctl);     \
    else hipLaunchKernelGGL((k_cg_update1<D, false>), dim3(g1), dim3(256), 0, s, nRows, dinv, u, w, p, sv, x, r, scal, it, ctl)
        if (dim == 3) { CALL1(3); } else if (dim == 2) { CALL1(2); } else { CALL1(1); }
#undef CALL1
        CHECK_LAUNCH();
        return;
    }
    const int grid = pair_grid(nRows * NR);
#define CALL(D)                                                                                                                          \
    if (skipU) hipLaunchKernelGGL((k_cg_update<D, true>), dim3(grid), dim3(256), 0, s, nRows, NR, dinv, u, w, p, sv, x, r, scal, it, ctl); \
    else hipLaunchKernelGGL((k_cg_update<D, false>), dim3(grid), dim3(256), 0, s, nRows, NR, dinv, u, w, p, sv, x, r, scal, it, ctl)
    if (dim == 3) { CALL(3); } else if (dim == 2) { CALL(2); } else { CALL(1); }
#undef CALL
    CHECK_LAUNCH();
}
void launch_cg_init(int dim, int64_t nRows, int NR, const double *dinv, const double *r, double *u, double *scal, bool skipU, hipStream_t s) {
    const int grid = pair_grid(nRows * NR);
#define CALL(D)                                                                                                  \
    if (skipU) hipLaunchKernelGGL((k_cg_init<D, true>), dim3(grid), dim3(256), 0, s, nRows, NR, dinv, r, u, scal); \
    else hipLaunchKernelGGL((k_cg_init<D, false>), dim3(grid), dim3(256), 0, s, nRows, NR, dinv, r, u, scal)
    if (dim == 3) { CALL(3); } else if (dim == 2) { CALL(2); } else { CALL(1); }
#undef CALL
    CHECK_LAUNCH();
}
void launch_tl_restrict_nr(const TLArgs &t, int NR, const int32_t *aggPtr, const int32_t *dofsByAgg, const double *w, double *rc, hipStream_t s) {
    if (t.nDoF < (int64_t)t.nAgg * 160 && ((t.dim == 3 && (NR == 2 || NR == 6)) || (t.dim == 2 && NR == 3))) {      // small aggregates: a wave each
        const unsigned grid = (unsigned)((t.nAgg + 3) / 4);
        if (t.dim == 3 && NR == 6) hipLaunchKernelGGL((k_tl_restrict_wave_nr<3, 6>), dim3(grid), dim3(256), 0, s, t, aggPtr, dofsByAgg, w, rc);
        else if (t.dim == 3) hipLaunchKernelGGL((k_tl_restrict_wave_nr<3, 2>), dim3(grid), dim3(256), 0, s, t, aggPtr, dofsByAgg, w, rc);
        else hipLaunchKernelGGL((k_tl_restrict_wave_nr<2, 3>), dim3(grid), dim3(256), 0, s, t, aggPtr, dofsByAgg, w, rc);
        CHECK_LAUNCH();
        return;
    }
    if (t.dim == 3) hipLaunchKernelGGL(k_tl_restrict_nr<3>, dim3(t.nAgg), dim3(256), 0, s, t, NR, aggPtr, dofsByAgg, w, rc);
    else hipLaunchKernelGGL(k_tl_restrict_nr<2>, dim3(t.nAgg), dim3(256), 0, s, t, NR, aggPtr, dofsByAgg, w, rc);
    CHECK_LAUNCH();
}
void launch_tl_gemv_nr(int64_t m, int64_t ld, int NR, const double *A, const double *x, double *y, hipStream_t s) {
    hipLaunchKernelGGL(k_tl_gemv_nr, dim3((unsigned)m), dim3(256), 0, s, m, ld, NR, A, x, y);
    CHECK_LAUNCH();
}
void launch_tl_apply_nr(const TLArgs &t, int NR, const double *dinv, const double *r, const double *yc, double *z, double *scal, int it,
                        const double *ctl, hipStream_t s) {
    const int grid = pair_grid(t.nDoF * NR);
    if (t.dim == 3) hipLaunchKernelGGL(k_tl_apply_nr<3>, dim3(grid), dim3(256), 0, s, t, NR, dinv, r, yc, z, scal, it, ctl);
    else hipLaunchKernelGGL(k_tl_apply_nr<2>, dim3(grid), dim3(256), 0, s, t, NR, dinv, r, yc, z, scal, it, ctl);
    CHECK_LAUNCH();
}
void launch_pack_rows(int64_t n, int W, const int32_t *idx, const double *src, double *dst, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_pack_rows, dim3(grid_for(n * W)), dim3(256), 0, s, n, W, idx, src, dst);
    CHECK_LAUNCH();
}
void launch_unpack_add_rows(int64_t n, int W, const int32_t *idx, const double *src, double *dst, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_unpack_add_rows, dim3(grid_for(n * W)), dim3(256), 0, s, n, W, idx, src, dst);
    CHECK_LAUNCH();
}
void launch_pack_rows_f32(int64_t n, int W, const int32_t *idx, const float *src, float *dst, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_pack_rows_f32, dim3(grid_for(n * W)), dim3(256), 0, s, n, W, idx, src, dst);
    CHECK_LAUNCH();
}
void launch_remap_i32(int64_t n, const int32_t *map, int32_t *v, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_remap_i32, dim3(grid_for(n)), dim3(256), 0, s, n, map, v);
    CHECK_LAUNCH();
}
void launch_interleave(int64_t nRows, int NR, int dim, const double *src, double *dst, bool toInterleaved, int64_t sepStride, hipStream_t s) {
    if (!nRows) return;
    hipLaunchKernelGGL(k_interleave, dim3(grid_for(nRows * NR * dim)), dim3(256), 0, s, nRows, NR, dim, src, dst, toInterleaved ? 1 : 0, sepStride);
    CHECK_LAUNCH();
}
void launch_norms_nr(int64_t nRows, int NR, int dim, const double *v, double *out, hipStream_t s) {
    if (!nRows) return;
    hipLaunchKernelGGL(k_norms_nr, dim3(grid_for(nRows)), dim3(256), 0, s, nRows, NR, dim, v, out);
    CHECK_LAUNCH();
}
void launch_mask_nr(int64_t nRows, int NR, int dim, const uint8_t *mask, double *v, hipStream_t s) {
    if (!nRows) return;
    hipLaunchKernelGGL(k_mask_nr, dim3(grid_for(nRows * NR * dim)), dim3(256), 0, s, nRows, NR, dim, mask, v);
    CHECK_LAUNCH();
}
void launch_scatter_values_nr(int64_t n, int NR, int dim, const int64_t *idx, const double *val, double *v, int64_t rowBound, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_scatter_values_nr, dim3(grid_for(n)), dim3(256), 0, s, n, NR, dim, idx, val, v, rowBound);
    CHECK_LAUNCH();
}


// ---- p-multigrid
// the gate's layout (GateScope, mfh_internal.hh): 0 = one classic loop, NR >= 1 = closed when all of NR lockstep loops are (histories t_gateStride apart)
thread_local int t_gateNr = 0;
thread_local int64_t t_gateStride = 0;
static MgGate mk_gate(const double *scal, int it, const double *stop) { return MgGate{scal, it, stop, t_gateNr, t_gateStride}; }
// kernels templated on (DIM, NR): NR = 1 and the batch sizes of op_batch_supported (3D: 2, 6; 2D: 3)
#define MG_NR_DISPATCH(dim, NR, K, grid, ...)                                                                     \
    do {                                                                                                          \
        if ((NR) == 1) {                                                                                          \
            if ((dim) == 3) hipLaunchKernelGGL((K<3, 1>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);              \
            else hipLaunchKernelGGL((K<2, 1>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);                         \
        } else if ((dim) == 3 && (NR) == 2) hipLaunchKernelGGL((K<3, 2>), dim3(grid), dim3(256), 0, s, __VA_ARGS__); \
        else if ((dim) == 3 && (NR) == 6) hipLaunchKernelGGL((K<3, 6>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);   \
        else if ((dim) == 2 && (NR) == 3) hipLaunchKernelGGL((K<2, 3>), dim3(grid), dim3(256), 0, s, __VA_ARGS__);   \
        else throw Error(MFH_ERR_UNSUPPORTED, "multigrid kernels: unsupported batch size");                      \
        CHECK_LAUNCH();                                                                                           \
    } while (0)
void launch_mg_cheb(int dim, int64_t nRows, const double *dinv, const double *rin, const double *t, double *rout, double *d, double *x,
                    double a, double b, bool first, bool assign, const double *scal, int it, const double *stop, hipStream_t s, int NR) {
    const MgGate g = mk_gate(scal, it, stop);
    const int grid = grid_for(nRows * NR, g_vecGridCap);
    if (dim == 1 && NR == 1) { hipLaunchKernelGGL(k_mg_cheb<1>, dim3(grid), dim3(256), 0, s, nRows, dinv, rin, t, rout, d, x, a, b, first ? 1 : 0, assign ? 1 : 0, g); CHECK_LAUNCH(); return; }
    MG_NR_DISPATCH(dim, NR, k_mg_cheb, grid, nRows, dinv, rin, t, rout, d, x, a, b, first ? 1 : 0, assign ? 1 : 0, g);
}
void launch_mg_restrict(int dim, int64_t nCoarse, const int32_t *fineOf, const int32_t *resPtr, const int32_t *resIdx, const double *r, const double *t,
                        const uint8_t *coarseMask, double *rc, const double *scal, int it, const double *stop, hipStream_t s, int NR, int64_t fineStride) {
    const MgGate g = mk_gate(scal, it, stop);
    const int grid = grid_for(nCoarse * MG_RESTRICT_LANES, MG_RESTRICT_LANES * g_vecGridCap);
    MG_NR_DISPATCH(dim, NR, k_mg_restrict, grid, nCoarse, fineOf, resPtr, resIdx, r, t, fineStride, coarseMask, rc, g);
}
void launch_mg_prolong_add(int dim, int64_t nFine, const int32_t *parA, const int32_t *parB, const double *xc, const uint8_t *fineMask, double *x,
                           const double *scal, int it, const double *stop, hipStream_t s, int ldc) {
    const MgGate g = mk_gate(scal, it, stop);
    if (ldc <= 0) ldc = dim;
    if (dim == 3) hipLaunchKernelGGL(k_mg_prolong_add<3>, dim3(grid_for(nFine, g_vecGridCap)), dim3(256), 0, s, nFine, parA, parB, xc, ldc, fineMask, x, g);
    else hipLaunchKernelGGL(k_mg_prolong_add<2>, dim3(grid_for(nFine, g_vecGridCap)), dim3(256), 0, s, nFine, parA, parB, xc, ldc, fineMask, x, g);
    CHECK_LAUNCH();
}
void launch_mg_prolong_add_nr(int dim, int NR, int64_t nFine, const int32_t *parA, const int32_t *parB, const double *xc, const uint8_t *fineMask, double *x,
                              int64_t vecStride, const double *scal, int64_t scalStride, int it, const double *stop, hipStream_t s) {
    const int grid = grid_for(nFine * NR, g_vecGridCap);
    if (dim == 3 && NR == 6) hipLaunchKernelGGL((k_mg_prolong_add_nr<3, 6>), dim3(grid), dim3(256), 0, s, nFine, parA, parB, xc, fineMask, x, vecStride, scal, scalStride, it, stop);
    else if (dim == 3 && NR == 2) hipLaunchKernelGGL((k_mg_prolong_add_nr<3, 2>), dim3(grid), dim3(256), 0, s, nFine, parA, parB, xc, fineMask, x, vecStride, scal, scalStride, it, stop);
    else if (dim == 2 && NR == 3) hipLaunchKernelGGL((k_mg_prolong_add_nr<2, 3>), dim3(grid), dim3(256), 0, s, nFine, parA, parB, xc, fineMask, x, vecStride, scal, scalStride, it, stop);
    else throw Error(MFH_ERR_UNSUPPORTED, "multigrid kernels: unsupported batch size");
    CHECK_LAUNCH();
}
void launch_mg_tl_prolong_add(const TLArgs &t, const double *yc, double *x, double alpha, const double *scal, int it, const double *stop, hipStream_t s, int NR) {
    const MgGate g = mk_gate(scal, it, stop);
    MG_NR_DISPATCH(t.dim, NR, k_mg_tl_prolong_add, grid_for(t.nDoF * NR, g_vecGridCap), t, yc, x, alpha, g);
}
void launch_fill_hash(int64_t n, double *v, hipStream_t s) {
    hipLaunchKernelGGL(k_fill_hash, dim3(grid_for(n, g_vecGridCap)), dim3(256), 0, s, n, v);
    CHECK_LAUNCH();
}
void launch_to_f32(int64_t n, const double *src, float *dst, hipStream_t s) {
    hipLaunchKernelGGL(k_to_f32, dim3(grid_for(n, g_vecGridCap)), dim3(256), 0, s, n, src, dst);
    CHECK_LAUNCH();
}
void launch_take_columns_i32(int64_t n, int W, int w, const int32_t *src, int32_t *dst, hipStream_t s) {
    hipLaunchKernelGGL(k_take_columns_i32, dim3(grid_for(n * w, g_vecGridCap)), dim3(256), 0, s, n, W, w, src, dst);
    CHECK_LAUNCH();
}
void launch_mg_diff(int64_t n, const double *a, const double *b, double *out, const double *scal, int it, const double *stop, hipStream_t s) {
    hipLaunchKernelGGL(k_mg_diff, dim3(grid_for(n, g_vecGridCap)), dim3(256), 0, s, n, a, b, out, mk_gate(scal, it, stop));
    CHECK_LAUNCH();
}
void launch_mg_rz(int64_t n, const double *r, double *z, const uint8_t *mask, double *scalOut, int it, const double *scal, const double *stop, hipStream_t s) {
    hipLaunchKernelGGL(k_mg_rz, dim3(grid_for(n)), dim3(256), 0, s, n, r, z, mask, scalOut, mk_gate(scal, it, stop), t_det);
    launch_det_finish(s);
    CHECK_LAUNCH();
}
// the V-cycle's last smoothing step and the PCG's r.z in one kernel (k_mg_cheb_rz)
void launch_mg_cheb_rz(int dim, int64_t nRows, const double *dinv, const float *dinv32, const double *rin, const double *t, double *x, double b, const uint8_t *mask,
                       double *scalOut, int it, const double *scal, const double *stop, hipStream_t s) {
    const int grid = grid_for(nRows / 2);
    if (dinv32) {
        if (dim == 3) hipLaunchKernelGGL((k_mg_cheb_rz<3, float>), dim3(grid), dim3(256), 0, s, nRows, dinv32, rin, t, x, b, mask, scalOut, mk_gate(scal, it, stop), t_det);
        else if (dim == 2) hipLaunchKernelGGL((k_mg_cheb_rz<2, float>), dim3(grid), dim3(256), 0, s, nRows, dinv32, rin, t, x, b, mask, scalOut, mk_gate(scal, it, stop), t_det);
        else hipLaunchKernelGGL((k_mg_cheb_rz<1, float>), dim3(grid), dim3(256), 0, s, nRows, dinv32, rin, t, x, b, mask, scalOut, mk_gate(scal, it, stop), t_det);
    } else
    if (dim == 3) hipLaunchKernelGGL(k_mg_cheb_rz<3>, dim3(grid), dim3(256), 0, s, nRows, dinv, rin, t, x, b, mask, scalOut, mk_gate(scal, it, stop), t_det);
    else if (dim == 2) hipLaunchKernelGGL(k_mg_cheb_rz<2>, dim3(grid), dim3(256), 0, s, nRows, dinv, rin, t, x, b, mask, scalOut, mk_gate(scal, it, stop), t_det);
    else hipLaunchKernelGGL(k_mg_cheb_rz<1>, dim3(grid), dim3(256), 0, s, nRows, dinv, rin, t, x, b, mask, scalOut, mk_gate(scal, it, stop), t_det);
    launch_det_finish(s);
    CHECK_LAUNCH();
}



// ---- aggregate (lattice-stencil) levels of the multigrid hierarchy
#define ST_DISPATCH(dim, K, grid, ...)                                                      \
    do {                                                                                    \
        if ((dim) == 3) hipLaunchKernelGGL(K<3>, dim3(grid), dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL(K<2>, dim3(grid), dim3(256), 0, s, __VA_ARGS__);            \
        CHECK_LAUNCH();                                                                     \
    } while (0)
void launch_st_spmv(int dim, int64_t nAgg, const int32_t *nbr, const double *A, const float *A32, const double *x, double *y, const double *scal, int it, const double *stop, hipStream_t s, int NR) {
    MG_NR_DISPATCH(dim, NR, k_st_spmv, grid_for(nAgg * 6 * NR), nAgg, nbr, A, A32, x, y, mk_gate(scal, it, stop));
}
void launch_st_dinv(int dim, int64_t nAgg, const double *A, double *Dinv, hipStream_t s) { ST_DISPATCH(dim, k_st_dinv, grid_for(nAgg), nAgg, A, Dinv); }
void launch_st_cheb(int dim, int64_t nAgg, const double *Dinv, const double *rin, const double *t, double *rout, double *d, double *x, double a, double b,
                    bool first, bool assign, const double *scal, int it, const double *stop, hipStream_t s, int NR) {
    MG_NR_DISPATCH(dim, NR, k_st_cheb, grid_for(nAgg * NR), nAgg, Dinv, rin, t, rout, d, x, a, b, first ? 1 : 0, assign ? 1 : 0, mk_gate(scal, it, stop));
}
void launch_st_rap(int dim, int64_t nParents, const int32_t *childPtr, const int32_t *childIdx, const int32_t *nbr, const double *A, const int32_t *parent,
                   const double *rel, const int32_t *coordC, double *Ac, const int *wrapNbC, hipStream_t s) {
    const int wx = wrapNbC ? wrapNbC[0] : 0, wy = wrapNbC ? wrapNbC[1] : 0, wz = wrapNbC ? wrapNbC[2] : 0;
    ST_DISPATCH(dim, k_st_rap, grid_for(nParents * 27), nParents, childPtr, childIdx, nbr, A, parent, rel, coordC, Ac, wx, wy, wz);
}
void launch_st_restrict(int dim, int64_t nParents, const int32_t *childPtr, const int32_t *childIdx, const double *rel, const double *r, const double *t, double *rc,
                        const double *scal, int it, const double *stop, hipStream_t s, int NR) {
    MG_NR_DISPATCH(dim, NR, k_st_restrict, grid_for(nParents * NR), nParents, childPtr, childIdx, rel, r, t, rc, mk_gate(scal, it, stop));
}
void launch_st_prolong_add(int dim, int64_t nAgg, const int32_t *parent, const double *rel, const double *xc, double *x, double alpha, const double *scal, int it,
                           const double *stop, hipStream_t s, int NR) {
    MG_NR_DISPATCH(dim, NR, k_st_prolong_add, grid_for(nAgg * NR), nAgg, parent, rel, xc, x, alpha, mk_gate(scal, it, stop));
}
void launch_st_to_dense(int dim, int64_t nAgg, const int32_t *nbr, const double *A, double *Ad, hipStream_t s) {
    ST_DISPATCH(dim, k_st_to_dense, grid_for(nAgg * 27 * 36), nAgg, nbr, A, Ad);
}
#undef ST_DISPATCH
#undef MG_NR_DISPATCH
void launch_mg_zero(int64_t n, double *v, const double *scal, int it, const double *stop, hipStream_t s) {
    hipLaunchKernelGGL(k_mg_zero, dim3(grid_for(n, g_vecGridCap)), dim3(256), 0, s, n, v, mk_gate(scal, it, stop));
    CHECK_LAUNCH();
}

}} // namespace mfh::k
