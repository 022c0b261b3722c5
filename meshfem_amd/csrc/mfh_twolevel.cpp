// Two-level preconditioner, host side: geometric aggregation of the DoFs into a coarse space of
// per-aggregate rigid-body modes, and the dense SPD inverse of the (small) coarse operator.
//
// The reference solves with CHOLMOD (SparseMatrices.hh:1984-2296); a Jacobi/block-Jacobi PCG needs
// O(1/h) iterations on the same system. The additive coarse correction
//     M^-1 = D^-1 + Z (Z^T K Z)^-1 Z^T,   Z = [rigid-body modes of every aggregate, masked on fixed DoFs]
// removes the smooth error components that cause that growth; everything per-iteration runs on the
// device (mfh_kernels_solver.hip: k_tl_*), only this once-per-system setup touches the host.
#include "mfh_internal.hh"
#include <cmath>
#include <unordered_map>

namespace mfh {

// ------------------------------------------------------------------------------------------------
// aggregation: uniform bins over the bounding box, ~targetNodes DoFs per bin
// ------------------------------------------------------------------------------------------------
void build_aggregates(int dim, int64_t nDoF, const std::vector<double> &dofPos, int targetNodes, Aggregates &A) {
    A = Aggregates();
    A.dim = dim;
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int64_t n = 0; n < nDoF; ++n)
        for (int a = 0; a < dim; ++a) {
            mn[a] = std::min(mn[a], dofPos[(size_t)n * dim + a]);
            mx[a] = std::max(mx[a], dofPos[(size_t)n * dim + a]);
        }
    double vol = 1;
    for (int a = 0; a < dim; ++a) vol *= std::max(mx[a] - mn[a], 1e-300);
    const double H = std::pow(vol * std::max(1, targetNodes) / (double)std::max<int64_t>(1, nDoF), 1.0 / dim);
    int nb[3] = {1, 1, 1};
    for (int a = 0; a < dim; ++a) nb[a] = std::max(1, (int)std::floor((mx[a] - mn[a]) / H + 0.5));
    A.H = H;
    auto binOf = [&](int64_t n, int *ib) {
        for (int a = 0; a < 3; ++a) ib[a] = 0;
        for (int a = 0; a < dim; ++a) {
            const double w = (mx[a] - mn[a]) / nb[a];
            int b = w > 0 ? (int)std::floor((dofPos[(size_t)n * dim + a] - mn[a]) / w) : 0;
            ib[a] = std::min(std::max(b, 0), nb[a] - 1);
        }
    };
    // compact numbering of the non-empty bins (in bin order: deterministic)
    std::vector<int32_t> binId((size_t)nb[0] * nb[1] * nb[2], -1);
    std::vector<int32_t> rawBin((size_t)nDoF);
    for (int64_t n = 0; n < nDoF; ++n) {
        int ib[3];
        binOf(n, ib);
        rawBin[n] = (ib[2] * nb[1] + ib[1]) * nb[0] + ib[0];
        binId[rawBin[n]] = 0;
    }
    int32_t nAgg = 0;
    for (auto &b : binId) if (b == 0) b = nAgg++;
    A.nAgg = nAgg;
    A.aggOfDof.resize((size_t)nDoF);
    for (int64_t n = 0; n < nDoF; ++n) A.aggOfDof[n] = binId[rawBin[n]];
    // centroids
    A.centroid.assign((size_t)nAgg * 3, 0.0);
    std::vector<int32_t> cnt((size_t)nAgg, 0);
    for (int64_t n = 0; n < nDoF; ++n) {
        const int32_t a = A.aggOfDof[n];
        ++cnt[a];
        for (int c = 0; c < dim; ++c) A.centroid[(size_t)a * 3 + c] += dofPos[(size_t)n * dim + c];
    }
    for (int32_t a = 0; a < nAgg; ++a)
        for (int c = 0; c < 3; ++c) A.centroid[(size_t)a * 3 + c] /= std::max(1, cnt[a]);
    // CSR of DoFs by aggregate
    A.aggPtr.assign((size_t)nAgg + 1, 0);
    for (int32_t a = 0; a < nAgg; ++a) A.aggPtr[a + 1] = A.aggPtr[a] + cnt[a];
    A.dofsByAgg.resize((size_t)nDoF);
    {
        std::vector<int32_t> cur(A.aggPtr.begin(), A.aggPtr.end() - 1);
        for (int64_t n = 0; n < nDoF; ++n) A.dofsByAgg[(size_t)cur[A.aggOfDof[n]]++] = (int32_t)n;
    }
    aggregate_lattice_tables(dim, nb, binId, A);
}

void aggregate_lattice_tables(int dim, const int nb[3], const std::vector<int32_t> &binId, Aggregates &A) {
    for (int a = 0; a < 3; ++a) A.nb[a] = nb[a];
    const int32_t nAgg = A.nAgg;
    // colours (3 x 3 x 3 on the bin lattice) and, per aggregate and colour, the unique aggregate of that
    // colour among its 3^dim lattice neighbours (K couples only DoFs of adjacent bins when H >> h)
    const int nColor = dim == 3 ? 27 : 9;
    A.nColor = nColor;
    A.colorOfAgg.assign((size_t)nAgg, 0);
    A.binCoord.assign((size_t)nAgg * 3, 0);
    A.nbrOfColor.assign((size_t)nAgg * nColor, -1);
    for (int iz = 0; iz < nb[2]; ++iz)
        for (int iy = 0; iy < nb[1]; ++iy)
            for (int ix = 0; ix < nb[0]; ++ix) {
                const int32_t a = binId[((size_t)iz * nb[1] + iy) * nb[0] + ix];
                if (a < 0) continue;
                A.colorOfAgg[a] = (ix % 3) + 3 * (iy % 3) + 9 * (dim == 3 ? iz % 3 : 0);
                A.binCoord[(size_t)a * 3] = ix; A.binCoord[(size_t)a * 3 + 1] = iy; A.binCoord[(size_t)a * 3 + 2] = iz;
                for (int dz = (dim == 3 ? -1 : 0); dz <= (dim == 3 ? 1 : 0); ++dz)
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            const int jx = ix + dx, jy = iy + dy, jz = iz + dz;
                            if (jx < 0 || jy < 0 || jz < 0 || jx >= nb[0] || jy >= nb[1] || jz >= nb[2]) continue;
                            const int32_t b = binId[((size_t)jz * nb[1] + jy) * nb[0] + jx];
                            if (b < 0) continue;
                            const int col = (jx % 3) + 3 * (jy % 3) + 9 * (dim == 3 ? jz % 3 : 0);
                            A.nbrOfColor[(size_t)a * nColor + col] = b;
                        }
            }
    // lattices with fewer than 3 bins along a direction alias colours across the whole direction: the
    // probing stays exact because such bins are mutual neighbours only once (dx in {-1,0,1} distinct)
    A.binsTooFew = (nb[0] < 3 && nb[0] > 1) || (nb[1] < 3 && nb[1] > 1) || (dim == 3 && nb[2] < 3 && nb[2] > 1);
}

// ------------------------------------------------------------------------------------------------
// dense SPD inverse (blocked Cholesky + triangular inverse + L^-T L^-1), threaded
// A: n x n row-major, symmetric, overwritten by its inverse. Returns false if not positive definite.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int TB = 96;   // tile edge

// C[mi x nj] += sign * A[mi x k] * B[k x nj]      (row-major; the j loop vectorises)
inline void gemm_nn(int mi, int nj, int k, double sign, const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc) {
    for (int i = 0; i < mi; ++i) {
        double *ci = C + i * ldc;
        for (int q = 0; q < k; ++q) {
            const double a = sign * A[i * lda + q];
            const double *bq = B + q * ldb;
            for (int j = 0; j < nj; ++j) ci[j] += a * bq[j];
        }
    }
}
// C[mi x nj] += sign * A[k x mi]^T * B[k x nj]
inline void gemm_tn(int mi, int nj, int k, double sign, const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc) {
    for (int q = 0; q < k; ++q) {
        const double *aq = A + q * lda, *bq = B + q * ldb;
        for (int i = 0; i < mi; ++i) {
            const double a = sign * aq[i];
            double *ci = C + i * ldc;
            for (int j = 0; j < nj; ++j) ci[j] += a * bq[j];
        }
    }
}
// C[mi x nj] += sign * A[mi x k] * B[nj x k]^T  via a local transpose of B
inline void gemm_nt(int mi, int nj, int k, double sign, const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc) {
    double Bt[TB * TB];
    for (int j = 0; j < nj; ++j)
        for (int q = 0; q < k; ++q) Bt[q * TB + j] = B[j * ldb + q];
    gemm_nn(mi, nj, k, sign, A, lda, Bt, TB, C, ldc);
}
} // namespace

bool spd_inverse_inplace(int64_t n, double *A) {
    if (n == 0) return true;
    const int64_t nt = (n + TB - 1) / TB;
    auto T = [&](double *M, int64_t bi, int64_t bj) { return M + bi * TB * n + bj * TB; };
    auto tsz = [&](int64_t b) { return (int)std::min<int64_t>(TB, n - b * TB); };
    bool ok = true;
    // ---- right-looking blocked Cholesky, L in the lower triangle of A
    for (int64_t k = 0; k < nt && ok; ++k) {
        const int kb = tsz(k);
        double *Akk = T(A, k, k);
        for (int j = 0; j < kb; ++j) {   // unblocked potrf of the diagonal tile
            double d = Akk[j * n + j];
            for (int q = 0; q < j; ++q) d -= Akk[j * n + q] * Akk[j * n + q];
            if (!(d > 0)) { ok = false; break; }
            d = std::sqrt(d);
            Akk[j * n + j] = d;
            for (int i = j + 1; i < kb; ++i) {
                double s = Akk[i * n + j];
                for (int q = 0; q < j; ++q) s -= Akk[i * n + q] * Akk[j * n + q];
                Akk[i * n + j] = s / d;
            }
        }
        if (!ok) break;
        // panel: L_ik = A_ik L_kk^-T
        parallel_ranges(nt - k - 1, [&](int64_t b, int64_t e, int) {
            for (int64_t ii = b; ii < e; ++ii) {
                const int64_t i = k + 1 + ii;
                const int ib = tsz(i);
                double *Aik = T(A, i, k);
                for (int r = 0; r < ib; ++r)
                    for (int j = 0; j < kb; ++j) {
                        double s = Aik[r * n + j];
                        for (int q = 0; q < j; ++q) s -= Aik[r * n + q] * Akk[j * n + q];
                        Aik[r * n + j] = s / Akk[j * n + j];
                    }
            }
        }, 1);
        // trailing update (lower triangle tiles): A_ij -= L_ik L_jk^T
        const int64_t rem = nt - k - 1;
        parallel_ranges(rem * (rem + 1) / 2, [&](int64_t b, int64_t e, int) {
            for (int64_t t = b; t < e; ++t) {
                int64_t i = (int64_t)((std::sqrt(8.0 * (double)t + 1) - 1) / 2);
                while ((i + 1) * (i + 2) / 2 <= t) ++i;
                while (i * (i + 1) / 2 > t) --i;
                const int64_t j = t - i * (i + 1) / 2;
                const int64_t bi = k + 1 + i, bj = k + 1 + j;
                gemm_nt(tsz(bi), tsz(bj), kb, -1.0, T(A, bi, k), n, T(A, bj, k), n, T(A, bi, bj), n);
            }
        }, 1);
    }
    if (!ok) return false;
    // ---- X = L^-1 (lower triangular), one block column per task:
    //      X_cc = L_cc^-1 ;  X_ic = -L_ii^-1 sum_{q=c}^{i-1} L_iq X_qc
    std::vector<double> X((size_t)n * n, 0.0);
    parallel_ranges(nt, [&](int64_t cb, int64_t ce, int) {
        std::vector<double> acc((size_t)TB * TB);
        for (int64_t c = cb; c < ce; ++c) {
            const int cw = tsz(c);
            for (int64_t i = c; i < nt; ++i) {
                const int ib = tsz(i);
                std::fill(acc.begin(), acc.end(), 0.0);
                if (i == c) for (int r = 0; r < ib; ++r) acc[(size_t)r * TB + r] = 1.0;
                for (int64_t q = c; q < i; ++q) gemm_nn(ib, cw, tsz(q), -1.0, T(A, i, q), n, T(X.data(), q, c), n, acc.data(), TB);
                // forward substitution with the diagonal tile L_ii
                const double *Lii = T(A, i, i);
                double *Xic = T(X.data(), i, c);
                for (int r = 0; r < ib; ++r) {
                    for (int q = 0; q < r; ++q) {
                        const double l = Lii[r * n + q];
                        for (int j = 0; j < cw; ++j) acc[(size_t)r * TB + j] -= l * acc[(size_t)q * TB + j];
                    }
                    const double inv = 1.0 / Lii[r * n + r];
                    for (int j = 0; j < cw; ++j) { acc[(size_t)r * TB + j] *= inv; Xic[r * n + j] = acc[(size_t)r * TB + j]; }
                }
            }
        }
    }, 1);
    // ---- A^-1 = X^T X: tile (I,J), J <= I:  sum_{Q >= I} X_QI^T X_QJ
    parallel_ranges(nt * (nt + 1) / 2, [&](int64_t b, int64_t e, int) {
        std::vector<double> acc((size_t)TB * TB);
        for (int64_t t = b; t < e; ++t) {
            int64_t I = (int64_t)((std::sqrt(8.0 * (double)t + 1) - 1) / 2);
            while ((I + 1) * (I + 2) / 2 <= t) ++I;
            while (I * (I + 1) / 2 > t) --I;
            const int64_t J = t - I * (I + 1) / 2;
            std::fill(acc.begin(), acc.end(), 0.0);
            for (int64_t Q = I; Q < nt; ++Q) gemm_tn(tsz(I), tsz(J), tsz(Q), 1.0, T(X.data(), Q, I), n, T(X.data(), Q, J), n, acc.data(), TB);
            double *out = T(A, I, J);
            for (int r = 0; r < tsz(I); ++r)
                for (int j = 0; j < tsz(J); ++j) out[r * n + j] = acc[(size_t)r * TB + j];
        }
    }, 1);
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = i + 1; j < n; ++j) A[(size_t)i * n + j] = A[(size_t)j * n + i];
    return true;
}

} // namespace mfh
