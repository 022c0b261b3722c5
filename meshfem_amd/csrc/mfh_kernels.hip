// HIP kernels for gfx950 (MI355X, CDNA4): element embedding, per-element stiffness blocks,
// owner-computes (gather) and atomic (scatter) assembly into a tiled block-CSR, block-CSR SpMV,
// and the fused PCG vector kernels.  All arithmetic is FP64 (the reference's Real = double,
// Types.hh:8); the path is HBM/LDS bound, so there is no MFMA here.
//
// Data layout of K values ("tiled BSR"): block slot s, component c (row-major in the dim x dim
// block) lives at vals[((s >> 6) * NB + c) * 64 + (s & 63)], NB = dim*dim. Consecutive lanes that
// own consecutive slots therefore read/write 512 contiguous bytes per component: every wave-wide
// load/store of K is a fully coalesced 8 B/lane access on both the assembly and the SpMV side.
#include "mfh_internal.hh"

namespace mfh { namespace k {

#define DEV __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
DEV int64_t tiled_index(int64_t slot, int c, int NB) { return ((slot >> 6) * NB + c) * 64 + (slot & 63); }

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

// XCD-aware work mapping. Workgroups are dispatched round-robin over the 8 XCDs (workgroup b runs on XCD b % 8) and
// every XCD has its own L2, so neighbouring work items (row chunks, element groups: they share element records, gather
// lists and x entries) should run on the SAME XCD: XCD x gets the x-th contiguous eighth of the n items.
// One workgroup per item: a bijection of [0, n).
DEV int64_t xcd_item(int64_t b, int64_t n) {
    const int64_t q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
    return x * q + (x < r ? x : r) + k;
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

// ------------------------------------------------------------------------------------------------
// One dim x dim block of the element stiffness matrix:
//   K_ij[c][d] = sum_q w_q sum_ab d_a phi_i(q) C_{acdb} d_b phi_j(q)      (LinearElasticity.hh:183-231)
// With grad phi_i(q) = alpha_i(q) u_a + beta_i(q) u_b the quadrature sum collapses onto the four
// precomputed pair coefficients S = vol * sum_q w_q {a_i a_j, a_i b_j, b_i a_j, b_i b_j}:
//   H[a][b] = sum_q w_q d_a phi_i d_b phi_j = u_a (S0 v_a + S1 v_b)^T + u_b (S2 v_a + S3 v_b)^T
// (same quadrature rule and points as the reference: GaussQuadrature.hh:115-127,283-295).
// ------------------------------------------------------------------------------------------------
// components per matrix entry: dim x dim blocks for elasticity, 1 for the scalar operators
template <int DIM, int MAT> DEV constexpr int mat_nb() { return (MAT == MAT_LAPLACE || MAT == MAT_MASS) ? 1 : DIM * DIM; }

template <int DIM, int DEG, int MAT, int ABL = 0>
DEV void elem_block(const double *__restrict__ g, const double *__restrict__ pairTab, const PairConst &pc, int i, int j, double *K) {
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    const double vol = (ABL & 2) ? 1.0 + i : g[12];
    if (MAT == MAT_MASS) {
        // int phi_i phi_j = vol * (reference value): exact for straight-sided simplices, equal to the
        // reference's Quadrature<K, 2 Deg> (exact for this integrand) up to rounding (MassMatrix.hh:66-77)
        K[0] = vol * pairTab[i * NPE + j];
        return;
    }
    double H[DIM][DIM];
    if (DEG == 1) {
        double gi[DIM], gj[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) { gi[a] = g[i * DIM + a]; gj[a] = g[j * DIM + a] * vol; }
#pragma unroll
        for (int a = 0; a < DIM; ++a)
#pragma unroll
            for (int b = 0; b < DIM; ++b) H[a][b] = gi[a] * gj[b];
    } else {
        const int si = sup_s<DIM, DEG>(i), ti = sup_t<DIM, DEG>(i), sj = sup_s<DIM, DEG>(j), tj = sup_t<DIM, DEG>(j);
        // Pair coefficients S = sum_q w_q coef_i(q) coef_j(q). Node i has the terms
        //   A: (4 lambda_l - o) grad lambda_{s_i}, l = s_i, o = 1 (vertex node) | l = t_i, o = 0 (edge node)
        //   B: (4 lambda_{s_i})  grad lambda_{t_i}, edge nodes only
        // and sum_q w_q (4 l_a - o)(4 l_b - o') takes only six values with the reference's rule (by
        // (o, o') class and a == b / a != b). They are read from the host-built quadrature table
        // (build_shape_tables), so no per-lane table loads are needed.
        constexpr int NV = DIM + 1;
        const bool vi = i < NV, vj = j < NV;
        const int lAi = vi ? si : ti, lBi = si, lAj = vj ? sj : tj, lBj = sj;
        // closed form of the table: S = c1 + [a==b] c2 - c3 (o + o') + o o'  with c1 = 16 m_ab (a != b),
        // c2 = 16 (m_aa - m_ab), c3 = 4 m_a (all three from the reference's quadrature rule). Pure
        // arithmetic on purpose: select chains over named constants get turned into a scratch table.
        const double c1 = pc.ee_ne, c2 = pc.ee_eq - pc.ee_ne, c3 = pc.ee_ne - pc.ve_ne;
        const double oi = vi ? 1.0 : 0.0, oj = vj ? 1.0 : 0.0;
        auto coef = [&](bool eq, double o, double o2) -> double { return (c1 + (eq ? c2 : 0.0)) - c3 * (o + o2) + o * o2; };
        double S0 = coef(lAi == lAj, oi, oj);                                    // A_i A_j
        double S1 = vj ? 0.0 : coef(lAi == lBj, oi, 0.0);                        // A_i B_j
        double S2 = vi ? 0.0 : coef(lBi == lAj, 0.0, oj);                        // B_i A_j
        double S3 = (vi | vj) ? 0.0 : coef(lBi == lBj, 0.0, 0.0);                // B_i B_j
        if (ABL & 1) { S0 = pairTab[0]; S1 = pairTab[1]; S2 = pairTab[2]; S3 = pairTab[3]; }
        S0 *= vol; S1 *= vol; S2 *= vol; S3 *= vol;
        double ua[DIM], ub[DIM], p[DIM], q[DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a) {
            ua[a] = (ABL & 2) ? (double)(si + a) : g[si * DIM + a]; ub[a] = (ABL & 2) ? (double)(ti - a) : g[ti * DIM + a];
            const double va = (ABL & 2) ? (double)(sj * a) : g[sj * DIM + a], vb = (ABL & 2) ? (double)(tj + 2 * a) : g[tj * DIM + a];
            p[a] = S0 * va + S1 * vb;
            q[a] = S2 * va + S3 * vb;
        }
#pragma unroll
        for (int a = 0; a < DIM; ++a)
#pragma unroll
            for (int b = 0; b < DIM; ++b) H[a][b] = ua[a] * p[b] + ub[a] * q[b];
    }
    if (MAT == MAT_LAPLACE) {
        // int grad phi_i . grad phi_j = tr(H)   (Laplacian.hh:38-48; Poisson.hh:33-38)
        double tr = 0;
#pragma unroll
        for (int a = 0; a < DIM; ++a) tr += H[a][a];
        K[0] = tr;
    } else if (MAT == MAT_ISO) {
        // C_acdb = lambda d_ac d_db + mu (d_ad d_cb + d_ab d_cd)  =>  K = lambda H + mu H^T + mu tr(H) I
        const double lam = (ABL & 2) ? 0.5 : g[13], mu = (ABL & 2) ? 0.25 : g[14];
        double tr = 0;
#pragma unroll
        for (int a = 0; a < DIM; ++a) tr += H[a][a];
#pragma unroll
        for (int c = 0; c < DIM; ++c)
#pragma unroll
            for (int d = 0; d < DIM; ++d) K[c * DIM + d] = lam * H[c][d] + mu * H[d][c] + (c == d ? mu * tr : 0.0);
    } else {
        // C_acdb = D(flat(a,c), flat(d,b))      (ElasticityTensor.hh:274-277)
        constexpr int ND = (DIM * (DIM + 1) / 2) * (DIM * (DIM + 1) / 2 + 1) / 2;
        double D[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) D[k] = g[13 + k];
#pragma unroll
        for (int c = 0; c < DIM; ++c)
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                double acc = 0;
#pragma unroll
                for (int a = 0; a < DIM; ++a)
#pragma unroll
                    for (int b = 0; b < DIM; ++b) acc += H[a][b] * D[dpack<DIM>(flat_idx<DIM>(a, c), flat_idx<DIM>(d, b))];
                K[c * DIM + d] = acc;
            }
    }
}

// ------------------------------------------------------------------------------------------------
// K1: element embedding + material record     (EmbeddedElement.hh:162-241, ElasticityTensor.hh:100-164)
// matMode: 0 const (lambda,mu) | 1 iso field (E[],nu[]) | 2 const general D (packed upper) |
//          3 orthotropic field (9 / 4 params per element) | 4 tensor field (flatLen^2 per element)
// ------------------------------------------------------------------------------------------------
template <int DIM>
__global__ void __launch_bounds__(256) k_geometry(int64_t nElem, const int32_t *__restrict__ elemNodes, int npe,
                                                  const double *__restrict__ vertPos, const double *__restrict__ mp,
                                                  int matMode, double *__restrict__ geo, int stride, int *negCount) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nElem) return;
    double P[DIM + 1][DIM];
#pragma unroll
    for (int k = 0; k <= DIM; ++k) {
        const int64_t v = elemNodes[e * npe + k];
#pragma unroll
        for (int a = 0; a < DIM; ++a) P[k][a] = vertPos[v * DIM + a];
    }
    double *g = geo + e * stride;
    double vol;
    if (DIM == 3) {
        // n0 = (p3-p1)x(p2-p1); 6V = (p0-p1).n0; gl0 = n0/6V; gl1 = (p2-p0)x(p3-p0)/6V; ...   (:223-230)
        auto cross = [](const double *a, const double *b, double *o) {
            o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
        };
        double d31[3], d21[3], d01[3], d20[3], d30[3], d10[3], n0[3], n1[3], n2[3], n3[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            d31[a] = P[3][a] - P[1][a]; d21[a] = P[2][a] - P[1][a]; d01[a] = P[0][a] - P[1][a];
            d20[a] = P[2][a] - P[0][a]; d30[a] = P[3][a] - P[0][a]; d10[a] = P[1][a] - P[0][a];
        }
        cross(d31, d21, n0); cross(d20, d30, n1); cross(d30, d10, n2); cross(d10, d20, n3);
        const double vol6 = d01[0] * n0[0] + d01[1] * n0[1] + d01[2] * n0[2];
        vol = vol6 / 6.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            g[0 + a] = n0[a] / vol6; g[3 + a] = n1[a] / vol6; g[6 + a] = n2[a] / vol6; g[9 + a] = n3[a] / vol6;
        }
    } else {
        // e0=p2-p1, e1=p0-p2, e2=p1-p0; 2A = e1.x e2.y - e1.y e2.x; gl_k = (-e_k.y, e_k.x)/2A   (:182-189)
        double E[3][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) { E[0][a] = P[2][a] - P[1][a]; E[1][a] = P[0][a] - P[2][a]; E[2][a] = P[1][a] - P[0][a]; }
        const double dA = E[1][0] * E[2][1] - E[1][1] * E[2][0];
        vol = dA / 2.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { g[k * 2 + 0] = -E[k][1] / dA; g[k * 2 + 1] = E[k][0] / dA; }
#pragma unroll
        for (int k = 6; k < 12; ++k) g[k] = 0.0;
    }
    g[12] = vol;
    if (!(vol >= 0)) atomicAdd(negCount, 1);
    constexpr int FL = DIM * (DIM + 1) / 2;
    constexpr int ND = FL * (FL + 1) / 2;
    if (matMode == 0) { g[13] = mp[0]; g[14] = mp[1]; }
    else if (matMode == 1) {
        const double E = mp[e], nu = mp[nElem + e];
        double lam = (nu * E) / ((1.0 + nu) * (1.0 - 2.0 * nu));
        if (DIM == 2) lam = (nu * E) / (1.0 - nu * nu);     // plane stress (ElasticityTensor.hh:108-112)
        g[13] = lam; g[14] = E / (2.0 + 2.0 * nu);
    } else if (matMode == 2) {
        for (int k = 0; k < ND; ++k) g[13 + k] = mp[k];
    } else if (matMode == 3) {
        for (int k = 0; k < ND; ++k) g[13 + k] = 0.0;
        if (DIM == 3) {
            const double *q = mp + e * 9;   // Ex,Ey,Ez,nuYX,nuZX,nuZY,muYZ,muZX,muXY  (:136-152)
            const double a00 = 1.0 / q[0], a01 = -q[3] / q[1], a02 = -q[4] / q[2], a11 = 1.0 / q[1], a12 = -q[5] / q[2],
                         a22 = 1.0 / q[2];
            const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
            const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
            const double det = a00 * c00 + a01 * c01 + a02 * c02;
            g[13 + dpack<3>(0, 0)] = c00 / det; g[13 + dpack<3>(0, 1)] = c01 / det; g[13 + dpack<3>(0, 2)] = c02 / det;
            g[13 + dpack<3>(1, 1)] = c11 / det; g[13 + dpack<3>(1, 2)] = c12 / det; g[13 + dpack<3>(2, 2)] = c22 / det;
            g[13 + dpack<3>(3, 3)] = q[6]; g[13 + dpack<3>(4, 4)] = q[7]; g[13 + dpack<3>(5, 5)] = q[8];
        } else {
            const double *q = mp + e * 4;   // Ex,Ey,nuYX,muXY                          (:154-164)
            const double a00 = 1.0 / q[0], a01 = -q[2] / q[1], a11 = 1.0 / q[1];
            const double det = a00 * a11 - a01 * a01;
            g[13 + dpack<2>(0, 0)] = a11 / det; g[13 + dpack<2>(0, 1)] = -a01 / det; g[13 + dpack<2>(1, 1)] = a00 / det;
            g[13 + dpack<2>(2, 2)] = q[3];
        }
    } else {
        const double *q = mp + e * FL * FL;
        for (int r = 0; r < FL; ++r)
            for (int c = r; c < FL; ++c) g[13 + dpack<DIM>(r, c)] = q[r * FL + c];
    }
}

// ------------------------------------------------------------------------------------------------
// K2+K3+K4 fused, owner-computes: one workgroup per row chunk (<= chunkSlots blocks of consecutive
// block rows). Every (element, i, j) contribution to those rows is computed by one lane and
// accumulated in LDS with ds_add_f64; the finished rows are written once, coalesced, with plain
// stores. No global atomics, no read-modify-write of K, no zero-fill pass.
// ------------------------------------------------------------------------------------------------
template <int DIM, int DEG, int MAT, int DBG = 0>
__global__ void __launch_bounds__(256) k_assemble_gather(AsmArgs a) {
    constexpr int NB = mat_nb<DIM, MAT>();
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    extern __shared__ __attribute__((aligned(16))) double acc[];   // [NB][chunkSlots + 2]
    const int CS = a.chunkSlots + 2;
    const PairConst pc{a.pairConst[0], a.pairConst[1], a.pairConst[2], a.pairConst[3], a.pairConst[4], a.pairConst[5]};
    const int64_t chunk = a.xcd ? xcd_item(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
    const int r0 = a.chunkRow[chunk], r1 = a.chunkRow[chunk + 1];
    const int s0 = a.rowPtr[r0];
    const int ns = a.rowPtr[r1] - s0;
    // LDS index = local slot + (s0 & 1): LDS pairs (2p, 2p+1) then coincide with 16-byte aligned
    // pairs of the tiled global layout and the write-out can use dwordx4 stores
    const int par = s0 & 1;
    for (int t = threadIdx.x; t < ns + par; t += 256)
#pragma unroll
        for (int c = 0; c < NB; ++c) acc[c * CS + t] = 0.0;
    __syncthreads();
    const int64_t kb = a.contribPtr[chunk], ke = a.contribPtr[chunk + 1];
    // U independent contributions per lane and trip: their index loads, element-record loads and
    // block arithmetic have no mutual dependence, so the loads of all U are in flight together
    // (the kernel is latency-bound: rocprof shows 65 % of wave cycles in s_waitcnt at U = 1).
    constexpr int U = (DBG >= 10 && DBG < 20) ? (DBG - 10) : 2;
    constexpr int ABL = (DBG >= 20 && DBG < 24) ? (DBG - 20) : 0;
    for (int64_t k0 = kb + threadIdx.x; k0 < ((DBG == 25 || DBG == 27) ? kb : ke); k0 += 256 * U) {
        uint32_t code[U];
        int ls[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t kk = k0 + (int64_t)u * 256;
            ok[u] = kk < ke;
            code[u] = ok[u] ? a.contribCode[kk] : a.contribCode[kb];
            ls[u] = (ok[u] ? (int)a.contribSlot[kk] : 0) + par;
        }
        double K[U][NB];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t e = code[u] / (NPE * NPE);
            const int ij = (int)(code[u] - e * (NPE * NPE));
            const int i = ij / NPE, j = ij - i * NPE;
            elem_block<DIM, DEG, MAT, ABL>(a.geo + (int64_t)e * a.geoStride, MAT == MAT_MASS ? a.massTable : a.pairTable, pc, i, j, K[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                if (DBG == 1) acc[c * CS + ls[u]] += K[u][c];                     // timing experiment only (racy)
                else if (DBG == 2) acc[c * CS + (threadIdx.x & (CS - 1))] = K[u][c]; // timing experiment only
                else unsafeAtomicAdd(&acc[c * CS + ls[u]], K[u][c]);
            }
        }
    }
    __syncthreads();
    if (DBG == 24 && ns > 1) return;       // timing experiment: no write-out
    // write-out: one 16-byte store per lane and component (two adjacent slots); the store path is
    // issue-bound, so halving the store-instruction count matters more than anything else here
    const int nl = ns + par;                               // LDS entries [par, nl) are live
    const int64_t sbase = (int64_t)s0 - par;               // even: LDS index t <-> global slot sbase + t
    const int pfirst = par;                                // first fully live pair (pair 0 holds a dead entry if par)
    const int plast = nl >> 1;                             // pairs [pfirst, plast) are fully live
    for (int p = pfirst + threadIdx.x; p < plast; p += 256) {
        const int t = 2 * p;
        const int64_t s = sbase + t;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            const double2 v = *reinterpret_cast<const double2 *>(&acc[c * CS + t]);
            double2 *dst = reinterpret_cast<double2 *>(&a.vals[tiled_index(s, c, NB)]);
            typedef double dv2 __attribute__((ext_vector_type(2)));
            // non-temporal: K is not re-read by this kernel; keeping it out of L2 leaves the cache to the
            // element records and gather lists (2.12 -> 1.71 ms at 1.5 M P2 tets)
            if (DBG != 28) { dv2 w = {v.x, v.y}; __builtin_nontemporal_store(w, reinterpret_cast<dv2 *>(dst)); }
            else *dst = v;
        }
    }
    // the (at most two) slots that are not part of a fully live pair
    if (threadIdx.x < 2) {
        const int t = threadIdx.x == 0 ? 1 : nl - 1;       // LDS entry 1 (if par) / last entry (if nl odd)
        const bool live = threadIdx.x == 0 ? (par == 1 && nl > 1) : ((nl & 1) && nl - 1 >= par && !(par == 1 && nl - 1 == 1));
        if (live) {
            const int64_t s = sbase + t;
#pragma unroll
            for (int c = 0; c < NB; ++c) a.vals[tiled_index(s, c, NB)] = acc[c * CS + t];
        }
    }
}

// Baseline variant: element-major, one lane per (element,i,j) block, global_atomic_add_f64 scatter
// into the (pre-zeroed) tiled values through the element->slot scatter map.
template <int DIM, int DEG, int MAT>
__global__ void __launch_bounds__(256) k_assemble_atomic(AsmArgs a) {
    constexpr int NB = mat_nb<DIM, MAT>();
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    const PairConst pc{a.pairConst[0], a.pairConst[1], a.pairConst[2], a.pairConst[3], a.pairConst[4], a.pairConst[5]};
    const int64_t total = a.nElem * (NPE * NPE);
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int32_t slot = a.scatterSlot[k];
        if (slot < 0) continue;
        const int64_t e = k / (NPE * NPE);
        const int ij = (int)(k - e * (NPE * NPE));
        const int i = ij / NPE, j = ij - i * NPE;
        double K[NB];
        elem_block<DIM, DEG, MAT>(a.geo + e * a.geoStride, MAT == MAT_MASS ? a.massTable : a.pairTable, pc, i, j, K);
#pragma unroll
        for (int c = 0; c < NB; ++c) unsafeAtomicAdd(&a.vals[tiled_index(slot, c, NB)], K[c]);
    }
}

// Dense per-element Ke (parity/debug): full (NPE*DIM)^2 row-major.
template <int DIM, int DEG, int MAT>
__global__ void __launch_bounds__(256) k_element_stiffness(AsmArgs a, int64_t first, int64_t count, double *out) {
    constexpr int NB = mat_nb<DIM, MAT>();
    constexpr int BS = NB == 1 ? 1 : DIM;          // block edge
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    constexpr int KS = NPE * BS;
    const PairConst pc{a.pairConst[0], a.pairConst[1], a.pairConst[2], a.pairConst[3], a.pairConst[4], a.pairConst[5]};
    const int64_t total = count * (NPE * NPE);
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t el = k / (NPE * NPE);
        const int ij = (int)(k - el * (NPE * NPE));
        const int i = ij / NPE, j = ij - i * NPE;
        double K[NB];
        elem_block<DIM, DEG, MAT>(a.geo + (first + el) * a.geoStride, MAT == MAT_MASS ? a.massTable : a.pairTable, pc, i, j, K);
        double *o = out + el * KS * KS;
#pragma unroll
        for (int c = 0; c < BS; ++c)
#pragma unroll
            for (int d = 0; d < BS; ++d) o[(i * BS + c) * KS + j * BS + d] = K[c * BS + d];
    }
}

// ------------------------------------------------------------------------------------------------
// K8: constantStrainLoad (LinearElasticity.hh:551-562, :135-162): l_i = (C_e : cstrain) . int grad phi_i,
// one lane per (element, node); int grad phi_i = vol (al_i gl[s_i] + be_i gl[t_i]) with al/be the
// integrals of the nodal coefficients (Interpolant::integrate, Functions.hh:246-253).
// K10: per-element averaged strain / stress (LinearElasticity.hh:99-123, :528-549).
// ------------------------------------------------------------------------------------------------
template <int DIM, int MAT>
DEV void elem_D_apply(const double *__restrict__ g, const double *sd /* shear-doubled flat strain */, double *out) {
    constexpr int FL = DIM * (DIM + 1) / 2;
    if (MAT == MAT_ISO) {
        const double lam = g[13], mu = g[14];
        double tr = 0;
#pragma unroll
        for (int a = 0; a < DIM; ++a) tr += sd[a];
#pragma unroll
        for (int a = 0; a < DIM; ++a) out[a] = lam * tr + 2 * mu * sd[a];
#pragma unroll
        for (int k = DIM; k < FL; ++k) out[k] = mu * sd[k];
    } else {
#pragma unroll
        for (int r = 0; r < FL; ++r) {
            double v = 0;
#pragma unroll
            for (int c = 0; c < FL; ++c) v += g[13 + dpack<DIM>(r, c)] * sd[c];
            out[r] = v;
        }
    }
}

template <int DIM>
DEV void load_corner_perturbation(const double *__restrict__ g, const int32_t *__restrict__ en, const double *__restrict__ deltaP,
                                  double (&gl)[DIM + 1][DIM], double (&dgl)[DIM + 1][DIM], double &relDeltaVol);

struct LoadArgs {
    int64_t nElem;
    int npe, geoStride;
    const double *geo;
    const int32_t *elemNodes;
    const int32_t *dofForNode;   // may be null
    double intGrad[20];          // npe x {al, be}
    double cstrain[6];           // flattened, TENSOR shear
};

template <int DIM, int DEG, int MAT>
__global__ void __launch_bounds__(256) k_constant_strain_load(LoadArgs a, const double *__restrict__ deltaP, double *__restrict__ out) {
    constexpr int FL = DIM * (DIM + 1) / 2;
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    const int64_t total = a.nElem * NPE;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (int64_t)gridDim.x * 256) {
        const int64_t e = k / NPE;
        const int i = (int)(k - e * NPE);
        const double *g = a.geo + e * a.geoStride;
        double sd[FL], cs[FL];
#pragma unroll
        for (int q = 0; q < FL; ++q) sd[q] = a.cstrain[q] * (q < DIM ? 1.0 : 2.0);   // shearDoubled (ElasticityTensor.hh:437-441)
        elem_D_apply<DIM, MAT>(g, sd, cs);
        const int si = sup_s<DIM, DEG>(i), ti = sup_t<DIM, DEG>(i);
        const double vol = g[12];
        // runtime-indexed kernel-argument array: read through a select chain over the small table
        double al = 0, be = 0;
#pragma unroll
        for (int q = 0; q < NPE; ++q) { al = (q == i) ? a.intGrad[2 * q] : al; be = (q == i) ? a.intGrad[2 * q + 1] : be; }
        double gi[DIM];
        if (deltaP) {
            // deltaConstantStrainLoad (LinearElasticity.hh:289-304, :1331-1348): delta (vol grad lambda) = vol (rel gl + dgl)
            double gl[DIM + 1][DIM], dgl[DIM + 1][DIM], rel;
            load_corner_perturbation<DIM>(g, a.elemNodes + e * NPE, deltaP, gl, dgl, rel);
            double gs[DIM], gt[DIM];
#pragma unroll
            for (int b = 0; b < DIM; ++b) { gs[b] = 0.0; gt[b] = 0.0; }
#pragma unroll
            for (int k2 = 0; k2 < DIM + 1; ++k2)
#pragma unroll
                for (int b = 0; b < DIM; ++b) {
                    const double v = rel * gl[k2][b] + dgl[k2][b];
                    gs[b] = (k2 == si) ? v : gs[b];
                    gt[b] = (k2 == ti) ? v : gt[b];
                }
#pragma unroll
            for (int b = 0; b < DIM; ++b) gi[b] = vol * (al * gs[b] + be * gt[b]);
        } else {
#pragma unroll
            for (int b = 0; b < DIM; ++b) gi[b] = vol * (al * g[si * DIM + b] + be * g[ti * DIM + b]);
        }
        int64_t dof = a.elemNodes[e * NPE + i];
        if (a.dofForNode) dof = a.dofForNode[dof];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            double v = 0;
#pragma unroll
            for (int b = 0; b < DIM; ++b) v += cs[flat_idx<DIM>(c, b)] * gi[b];
            unsafeAtomicAdd(&out[dof * DIM + c], v);
        }
    }
}

template <int DIM, int DEG, int MAT>
__global__ void __launch_bounds__(256) k_average_strain(LoadArgs a, const double *__restrict__ uNodes, double *__restrict__ out,
                                                        int wantStress, const double *__restrict__ uFixed,
                                                        const double *__restrict__ deltaP) {
    constexpr int FL = DIM * (DIM + 1) / 2;
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.nElem; e += (int64_t)gridDim.x * 256) {
        const double *g = a.geo + e * a.geoStride;
        double eps[DIM][DIM];
#pragma unroll
        for (int x = 0; x < DIM; ++x)
#pragma unroll
            for (int y = 0; y < DIM; ++y) eps[x][y] = 0.0;
#pragma unroll
        for (int i = 0; i < NPE; ++i) {
            const int si = sup_s<DIM, DEG>(i), ti = sup_t<DIM, DEG>(i);
            const double al = a.intGrad[2 * i], be = a.intGrad[2 * i + 1];
            const int64_t node = a.elemNodes[e * NPE + i];
            double gb[DIM], ui[DIM];
#pragma unroll
            for (int b = 0; b < DIM; ++b) { gb[b] = al * g[si * DIM + b] + be * g[ti * DIM + b]; ui[b] = uNodes[node * DIM + b]; }
#pragma unroll
            for (int x = 0; x < DIM; ++x)
#pragma unroll
                for (int y = 0; y < DIM; ++y) eps[x][y] += 0.5 * (ui[x] * gb[y] + ui[y] * gb[x]);   // LinearElasticity.hh:99-115
        }
        if (deltaP) {
            // deltaAverageStrainField (LinearElasticity.hh:1364-1374): + (delta strain)(uFixed), the strain of the fixed
            // field on the perturbed gradients (:259-277)
            double gl[DIM + 1][DIM], dgl[DIM + 1][DIM], rel;
            load_corner_perturbation<DIM>(g, a.elemNodes + e * NPE, deltaP, gl, dgl, rel);
#pragma unroll
            for (int i = 0; i < NPE; ++i) {
                const int si = sup_s<DIM, DEG>(i), ti = sup_t<DIM, DEG>(i);
                const double al = a.intGrad[2 * i], be = a.intGrad[2 * i + 1];
                const int64_t node = a.elemNodes[e * NPE + i];
                double gb[DIM], ui[DIM];
#pragma unroll
                for (int b = 0; b < DIM; ++b) { gb[b] = al * dgl[si][b] + be * dgl[ti][b]; ui[b] = uFixed[node * DIM + b]; }
#pragma unroll
                for (int x = 0; x < DIM; ++x)
#pragma unroll
                    for (int y = 0; y < DIM; ++y) eps[x][y] += 0.5 * (ui[x] * gb[y] + ui[y] * gb[x]);
            }
        }
        double ef[FL];
#pragma unroll
        for (int x = 0; x < DIM; ++x)
#pragma unroll
            for (int y = x; y < DIM; ++y) ef[flat_idx<DIM>(x, y)] = eps[x][y];
        if (wantStress) {
            double sd[FL], sg[FL];
#pragma unroll
            for (int q = 0; q < FL; ++q) sd[q] = ef[q] * (q < DIM ? 1.0 : 2.0);
            elem_D_apply<DIM, MAT>(g, sd, sg);
#pragma unroll
            for (int q = 0; q < FL; ++q) out[e * FL + q] = sg[q];
        } else {
#pragma unroll
            for (int q = 0; q < FL; ++q) out[e * FL + q] = ef[q];
        }
    }
}

// Average gradient of a scalar nodal field per element (PoissonMesh::gradUAverage, Poisson.hh:121-131):
// (1/vol) int sum_i u_i grad phi_i = sum_i u_i (al_i gl[s_i] + be_i gl[t_i]).
template <int DIM, int DEG>
__global__ void __launch_bounds__(256) k_average_gradient(LoadArgs a, const double *__restrict__ uNodes, double *__restrict__ out) {
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.nElem; e += (int64_t)gridDim.x * 256) {
        const double *g = a.geo + e * a.geoStride;
        double gr[DIM];
#pragma unroll
        for (int b = 0; b < DIM; ++b) gr[b] = 0.0;
#pragma unroll
        for (int i = 0; i < NPE; ++i) {
            const int si = sup_s<DIM, DEG>(i), ti = sup_t<DIM, DEG>(i);
            const double al = a.intGrad[2 * i], be = a.intGrad[2 * i + 1];
            const double ui = uNodes[a.elemNodes[e * NPE + i]];
#pragma unroll
            for (int b = 0; b < DIM; ++b) gr[b] += ui * (al * g[si * DIM + b] + be * g[ti * DIM + b]);
        }
#pragma unroll
        for (int b = 0; b < DIM; ++b) out[e * DIM + b] = gr[b];
    }
}

// ------------------------------------------------------------------------------------------------
// Block-CSR SpMV over row chunks (persistent workgroups, grid-stride over chunks).
// Phase 1: lane per block: coalesced loads of the NB components + column, gather x, block product
//          -> LDS partials.  Phase 2: lane per scalar row sums its partials.
// Optional: zero rows of fixed variables; accumulate dot(x_rows, y) into *dotOut (one atomic per WG).
// ------------------------------------------------------------------------------------------------
template <int DIM, bool PCG>
__global__ void __launch_bounds__(256) k_spmv(SpmvArgs a, const double *__restrict__ x, double *__restrict__ y,
                                             double *dotOut, double *scal, int it, const double *stopPtr) {
    constexpr int NB = DIM * DIM;
    extern __shared__ __attribute__((aligned(16))) double part[];  // [DIM][chunkSlots] + 16
    const int CS = a.chunkSlots;
    double *red = part + DIM * CS;
    if (PCG) {
        it += (int)stopPtr[3];   // iteration base of the current graph launch (0 outside graphs)
        // converged: every kernel of the remaining iterations is a no-op
        if (scal[(int64_t)it * 4 + 2] <= stopPtr[0]) return;
        dotOut = scal + (int64_t)it * 4 + 1;
    }
    double dot = 0.0;
    int64_t chunkFirst = blockIdx.x, chunkEnd = a.nChunk, chunkStride = gridDim.x;
    if (a.xcd) xcd_span(a.nChunk, chunkFirst, chunkEnd, chunkStride);
    for (int64_t chunk = chunkFirst; chunk < chunkEnd; chunk += chunkStride) {
        const int r0 = a.chunkRow[chunk], r1 = a.chunkRow[chunk + 1];
        const int s0 = a.rowPtr[r0];
        const int ns = a.rowPtr[r1] - s0;
        for (int t = threadIdx.x; t < ns; t += 256) {
            const int64_t s = (int64_t)s0 + t;
            const int64_t col = a.colIdx[s];
            double xv[DIM], A[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c) A[c] = a.vals[tiled_index(s, c, NB)];
#pragma unroll
            for (int d = 0; d < DIM; ++d) xv[d] = x[col * DIM + d];
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                double v = 0;
#pragma unroll
                for (int d = 0; d < DIM; ++d) v += A[c * DIM + d] * xv[d];
                part[c * CS + t] = v;
            }
        }
        __syncthreads();
        const int nscalar = (r1 - r0) * DIM;
        for (int idx = threadIdx.x; idx < nscalar; idx += 256) {
            const int rl = idx / DIM, c = idx - rl * DIM;
            const int64_t r = r0 + rl;
            const int b = a.rowPtr[r] - s0, e = a.rowPtr[r + 1] - s0;
            double v = 0;
            for (int t = b; t < e; ++t) v += part[c * CS + t];
            const int64_t gi = r * DIM + c;
            if (a.fixedMask && a.fixedMask[gi]) v = 0.0;
            y[gi] = v;
            if (dotOut) dot += v * x[gi];
        }
        __syncthreads();
    }
    if (dotOut) {
        double v[1] = {dot};
        block_sum<1>(v, red);
        if (threadIdx.x == 0) unsafeAtomicAdd(dotOut, v[0]);
    }
}

// ------------------------------------------------------------------------------------------------
// Matrix-free operator y = K x: one lane per (element, local node i) pair evaluates the row i of the
// element matrix block by block in registers (same elem_block as the assembly: identical values) and
// applies it to the gathered x; pairs of a row chunk are reduced in LDS. Trades the 72 B/block of the
// assembled SpMV for ~73 FP64 flops/block: HBM traffic drops from nnzb*76 B to the element records +
// pair lists + x, the FP64 VALU (idle in the assembled SpMV) does the work.
// ------------------------------------------------------------------------------------------------
template <int DIM, int DEG, int MAT, bool PCG, int UNR = 0>
__global__ void __launch_bounds__(256) k_spmv_mf(SpmvMfArgs a, const double *__restrict__ x, double *__restrict__ y, double *dotOut,
                                                 double *scal, int it, const double *stopPtr) {
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    constexpr int NB = mat_nb<DIM, MAT>();
    constexpr int BS = NB == 1 ? 1 : DIM;
    extern __shared__ __attribute__((aligned(16))) double mfacc[];   // [maxRows * BS] + 16
    double *red = mfacc + a.maxRows * BS;
    if (PCG) {
        it += (int)stopPtr[3];
        if (scal[(int64_t)it * 4 + 2] <= stopPtr[0]) return;
        dotOut = scal + (int64_t)it * 4 + 1;
    }
    const PairConst pc{a.pairConst[0], a.pairConst[1], a.pairConst[2], a.pairConst[3], a.pairConst[4], a.pairConst[5]};
    const double *tab = MAT == MAT_MASS ? a.massTable : a.pairTable;
    double dot = 0.0;
    int64_t chunkFirst = blockIdx.x, chunkEnd = a.nChunk, chunkStride = gridDim.x;
    if (a.xcd) xcd_span(a.nChunk, chunkFirst, chunkEnd, chunkStride);
    for (int64_t chunk = chunkFirst; chunk < chunkEnd; chunk += chunkStride) {
        const int r0 = a.chunkRow[chunk], r1 = a.chunkRow[chunk + 1];
        const int nr = (r1 - r0) * BS;
        for (int t = threadIdx.x; t < nr; t += 256) mfacc[t] = 0.0;
        __syncthreads();
        const int64_t kb = a.pairPtr[chunk], ke = a.pairPtr[chunk + 1];
        for (int64_t k = kb + threadIdx.x; k < ke; k += 256) {
            const uint32_t code = a.pairCode[k];
            const int lr = a.pairRow[k];
            const uint32_t e = code / NPE;
            const int i = (int)(code - e * NPE);
            const double *g = a.geo + (int64_t)e * a.geoStride;
            const int32_t *en = a.elemNodes + (int64_t)e * NPE;
            double out[BS];
#pragma unroll
            for (int c = 0; c < BS; ++c) out[c] = 0.0;
            constexpr int UNROLL_PAIRS = UNR == 0 ? NPE : UNR;
#pragma unroll UNROLL_PAIRS
            for (int j = 0; j < NPE; ++j) {
                int64_t col = en[j];
                if (a.dofForNode) col = a.dofForNode[col];
                double xv[BS], K[NB];
#pragma unroll
                for (int d = 0; d < BS; ++d) xv[d] = x[col * BS + d];
                elem_block<DIM, DEG, MAT>(g, tab, pc, i, j, K);
#pragma unroll
                for (int c = 0; c < BS; ++c)
#pragma unroll
                    for (int d = 0; d < BS; ++d) out[c] += K[c * BS + d] * xv[d];
            }
#pragma unroll
            for (int c = 0; c < BS; ++c) unsafeAtomicAdd(&mfacc[lr * BS + c], out[c]);
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < nr; idx += 256) {
            const int64_t gi = (int64_t)r0 * BS + idx;
            double v = mfacc[idx];
            if (a.fixedMask && a.fixedMask[gi]) v = 0.0;
            y[gi] = v;
            if (dotOut) dot += v * x[gi];
        }
        __syncthreads();
    }
    if (dotOut) {
        double v[1] = {dot};
        block_sum<1>(v, red);
        if (threadIdx.x == 0) unsafeAtomicAdd(dotOut, v[0]);
    }
}

// ------------------------------------------------------------------------------------------------
// Matrix-free ELASTICITY operator in two passes -- the stresses are shared by the npe rows of an element
// instead of being recomputed per (element, node) pair:
//   k_mf_forces: one lane per element. With the reference's quadrature rule every point q belongs to a vertex
//     (lambda_k(q) = c0 if k == q else c1), so grad u(q) = G_b + (a-b) x_q (x) gl_q + (A-B) sum_{k != q} x_edge(k,q) (x) gl_k
//     with G_b = sum_k (b x_k + B sum_{edges m at k} x_m) (x) gl_k,  a = 4 c0 - 1, b = 4 c1 - 1, A = 4 c0, B = 4 c1
//     (EmbeddedElement.hh:288-313). With S' = vol w sum_q sigma_q and R_q = vol w (a-b) sigma_q the nodal forces
//     f_i = int sigma(u) grad phi_i are
//     vertex k: (b S' + R_k) gl_k ;  edge (s,t): (B S' + R_t) gl_s + (B S' + R_s) gl_t ;  P1: vol sigma gl_i
//     (same quadrature as the assembled K: identical up to rounding), written element-major: 240 B per P2 tet.
//   k_mf_rows: one lane per (element, node) pair (lists of build_mf_lists_device): y_row = sum of the pairs' forces
//     -- a pure gather-sum through LDS, no arithmetic.
// HBM traffic: records + connectivity + x + 2 x 240 B/element of forces + pair lists ~ 4.5 GB at 5.2 M P2 tets,
// against 15.7 GB for the assembled SpMV.
// ------------------------------------------------------------------------------------------------
template <int DIM> DEV constexpr int edge_between(int k, int q) {
    // local edge index joining vertices k and q (Simplex.hh:43-47): (0,1),(1,2),(2,0),(0,3),(2,3),(1,3)
    const int lo = k < q ? k : q, hi = k < q ? q : k;
    if (lo == 0 && hi == 1) return 0;
    if (lo == 1 && hi == 2) return 1;
    if (lo == 0 && hi == 2) return 2;
    if (lo == 0 && hi == 3) return 3;
    if (lo == 2 && hi == 3) return 4;
    return 5;   // (1,3)
}

// nodal forces f_i = int sigma(u) grad phi_i of element e for the nodal vectors gathered from x
template <int DIM, int DEG, int MAT, class Emit>
DEV void elem_forces_core(const SpmvMfArgs &a, int64_t e, const double (&xl)[(DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6)][DIM],
                          const Emit &emit);

template <int DIM, int DEG, int MAT>
DEV void elem_forces(const SpmvMfArgs &a, int64_t e, const double *__restrict__ x,
                     double (&f)[(DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6)][DIM]) {
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    const int32_t *en = a.elemNodes + e * NPE;
    double xl[NPE][DIM];
#pragma unroll
    for (int j = 0; j < NPE; ++j) {
        int64_t col = en[j];
        if (a.dofForNode) col = a.dofForNode[col];
#pragma unroll
        for (int d = 0; d < DIM; ++d) xl[j][d] = x[col * DIM + d];
    }
    elem_forces_core<DIM, DEG, MAT>(a, e, xl, [&](int j, const double *fv) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) f[j][d] = fv[d];
    });
}

// The arithmetic of elem_forces on nodal vectors that are already gathered, as a bilinear form in the barycentric
// gradients: the strain of u is built with glS, the test functions' gradients with glT,
//     f_i = vol int sigma(u; glS) grad phi_i(glT).
// The operator itself is glS == glT == grad lambda; the discrete shape derivative (k_apply_delta_K) feeds perturbed
// gradients into either slot. Every nodal force is handed to `emit(j, f_j)` as soon as it is complete (the cluster
// kernel adds it to LDS right away instead of keeping 30 values live).
template <int DIM, int DEG, int MAT, class Emit>
DEV void elem_forces_bilinear(const double *__restrict__ g /* element record: material */, double vol,
                              const double (&gl)[DIM + 1][DIM], const double (&glT)[DIM + 1][DIM],
                              const double (&xl)[(DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6)][DIM], const Emit &emit) {
    constexpr int NV = DIM + 1;
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    constexpr int FL = DIM * (DIM + 1) / 2;
    auto stress_of = [&](const double (&G)[DIM][DIM], double *out) {
        double sd[FL];
#pragma unroll
        for (int p = 0; p < DIM; ++p)
#pragma unroll
            for (int q2 = p; q2 < DIM; ++q2)
                sd[flat_idx<DIM>(p, q2)] = (p == q2) ? G[p][p] : (G[p][q2] + G[q2][p]);   // shear-doubled strain
        elem_D_apply<DIM, MAT>(g, sd, out);
    };
    auto symv = [&](const double *T, const double *v, double *o) {   // o += T v  (T flat symmetric)
#pragma unroll
        for (int p = 0; p < DIM; ++p)
#pragma unroll
            for (int q2 = 0; q2 < DIM; ++q2) o[p] += T[flat_idx<DIM>(p, q2)] * v[q2];
    };
    if (DEG == 1) {
        double G[DIM][DIM];
#pragma unroll
        for (int p = 0; p < DIM; ++p)
#pragma unroll
            for (int q2 = 0; q2 < DIM; ++q2) {
                double v = 0;
#pragma unroll
                for (int j = 0; j < NV; ++j) v += xl[j][p] * gl[j][q2];
                G[p][q2] = v;
            }
        double sg[FL];
        stress_of(G, sg);
#pragma unroll
        for (int c = 0; c < FL; ++c) sg[c] *= vol;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            double fv[DIM] = {};
            symv(sg, glT[j], fv);
            emit(j, fv);
        }
    } else {
        constexpr double c0 = DIM == 3 ? 0.58541019662496845446 : 2.0 / 3.0;     // GaussQuadrature.hh:283-295 / :115-127
        constexpr double c1 = DIM == 3 ? 0.13819660112501051518 : 1.0 / 6.0;
        constexpr double b_ = 4 * c1 - 1, A_ = 4 * c0, B_ = 4 * c1, dAB = A_ - B_;   // a - b = A - B
        constexpr double wq = 1.0 / NV;
        double Gb[DIM][DIM];
#pragma unroll
        for (int p = 0; p < DIM; ++p)
#pragma unroll
            for (int q2 = 0; q2 < DIM; ++q2) Gb[p][q2] = 0.0;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double base[DIM];
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                double sum = 0;
#pragma unroll
                for (int o = 0; o < NV; ++o)
                    if (o != k) sum += xl[NV + edge_between<DIM>(k, o)][d];
                base[d] = b_ * xl[k][d] + B_ * sum;
            }
#pragma unroll
            for (int p = 0; p < DIM; ++p)
#pragma unroll
                for (int q2 = 0; q2 < DIM; ++q2) Gb[p][q2] += base[p] * gl[k][q2];
        }
        double S[FL], R[NV][FL];
#pragma unroll
        for (int c = 0; c < FL; ++c) S[c] = 0.0;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            double G[DIM][DIM];
#pragma unroll
            for (int p = 0; p < DIM; ++p)
#pragma unroll
                for (int q2 = 0; q2 < DIM; ++q2) {
                    double v = Gb[p][q2] + dAB * xl[q][p] * gl[q][q2];
#pragma unroll
                    for (int k = 0; k < NV; ++k)
                        if (k != q) v += dAB * xl[NV + edge_between<DIM>(k, q)][p] * gl[k][q2];
                    G[p][q2] = v;
                }
            double sg[FL];
            stress_of(G, sg);
#pragma unroll
            for (int c = 0; c < FL; ++c) { S[c] += sg[c]; R[q][c] = vol * wq * dAB * sg[c]; }
        }
        // vertex k: f = (b S' + R_k) gl_k ;  edge (s,t): f = (B S' + R_t) gl_s + (B S' + R_s) gl_t
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double T[FL];
#pragma unroll
            for (int c = 0; c < FL; ++c) T[c] = (vol * wq * b_) * S[c] + R[k][c];
            double fv[DIM] = {};
            symv(T, glT[k], fv);
            emit(k, fv);
        }
#pragma unroll
        for (int c = 0; c < FL; ++c) S[c] *= vol * wq * B_;
#pragma unroll
        for (int m = 0; m < NPE - NV; ++m) {
            const int sI = sup_s<DIM, DEG>(NV + m), tI = sup_t<DIM, DEG>(NV + m);
            double T1[FL], T2[FL];
#pragma unroll
            for (int c = 0; c < FL; ++c) { T1[c] = S[c] + R[tI][c]; T2[c] = S[c] + R[sI][c]; }
            double fv[DIM] = {};
            symv(T1, glT[sI], fv);
            symv(T2, glT[tI], fv);
            emit(NV + m, fv);
        }
    }
}

template <int DIM, int DEG, int MAT, class Emit>
DEV void elem_forces_core(const SpmvMfArgs &a, int64_t e, const double (&xl)[(DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6)][DIM],
                          const Emit &emit) {
    constexpr int NV = DIM + 1;
    const double *g = a.geo + e * a.geoStride;
    double gl[NV][DIM];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int d = 0; d < DIM; ++d) gl[k][d] = g[k * DIM + d];
    elem_forces_bilinear<DIM, DEG, MAT>(g, g[12], gl, gl, xl, emit);
}

// ------------------------------------------------------------------------------------------------
// Discrete shape derivatives (forward mode): change of the element quantities under a perturbation delta_p of the
// mesh vertices, with the nodal fields held fixed (LinearElasticity.hh:238-330; Simulator level :1301-1374).
// Everything depends on the geometry through (grad lambda, vol) only:
//     delta grad lambda_i = - sum_k grad lambda_k (grad lambda_i . delta_p_k)      (EmbeddedElement.hh:269-278)
//     delta vol / vol     =   sum_k grad lambda_k . delta_p_k                       (EmbeddedElement.hh:366-372)
// and grad phi_i has the same (s,t) support form in delta grad lambda as in grad lambda (:338-363), so each
// derivative is the original kernel evaluated on perturbed gradients (product rule on the bilinear forms).
// ------------------------------------------------------------------------------------------------
template <int DIM>
DEV void load_corner_perturbation(const double *__restrict__ g, const int32_t *__restrict__ en /* element's nodes: corners first */,
                                  const double *__restrict__ deltaP, double (&gl)[DIM + 1][DIM], double (&dgl)[DIM + 1][DIM],
                                  double &relDeltaVol) {
    constexpr int NV = DIM + 1;
    double dp[NV][DIM];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int64_t v = en[k];
#pragma unroll
        for (int d = 0; d < DIM; ++d) { gl[k][d] = g[k * DIM + d]; dp[k][d] = deltaP[v * DIM + d]; dgl[k][d] = 0.0; }
    }
    relDeltaVol = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) relDeltaVol += gl[k][d] * dp[k][d];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double sdot = 0;
#pragma unroll
            for (int d = 0; d < DIM; ++d) sdot += gl[i][d] * dp[k][d];
#pragma unroll
            for (int d = 0; d < DIM; ++d) dgl[i][d] -= gl[k][d] * sdot;
        }
    }
}

// flattened (tensor-shear) symmetrised gradient of the nodal field xl at quadrature point q, built on the gradients gl:
// sym(sum_i u_i (x) grad phi_i(x_q)); P2: grad phi_k = (4 lam_k - 1) gl_k, grad phi_(s,t) = 4 (lam_t gl_s + lam_s gl_t)
// (EmbeddedElement.hh:315-332) at the vertex-associated points lam_k(x_q) = c0 if k == q else c1.
// AT_VERTEX: evaluate at vertex q instead (lam_k = [k == q]), the nodal values of the degree-(DEG-1) strain interpolant
template <int DIM, int DEG, bool AT_VERTEX = false>
DEV void grad_u_at(const double (&xl)[(DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6)][DIM], const double (&gl)[DIM + 1][DIM],
                   int q, double (&G)[DIM][DIM]) {
    constexpr int NV = DIM + 1;
    constexpr double c0 = AT_VERTEX ? 1.0 : (DIM == 3 ? 0.58541019662496845446 : 2.0 / 3.0);
    constexpr double c1 = AT_VERTEX ? 0.0 : (DIM == 3 ? 0.13819660112501051518 : 1.0 / 6.0);
#pragma unroll
    for (int p = 0; p < DIM; ++p)
#pragma unroll
        for (int r = 0; r < DIM; ++r) G[p][r] = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double ck[DIM];
        if (DEG == 1) {
#pragma unroll
            for (int p = 0; p < DIM; ++p) ck[p] = xl[k][p];
        } else {
            const double lk = (k == q) ? c0 : c1;
#pragma unroll
            for (int p = 0; p < DIM; ++p) ck[p] = (4 * lk - 1) * xl[k][p];
#pragma unroll
            for (int o = 0; o < NV; ++o)
                if (o != k) {
                    const double lo = (o == q) ? c0 : c1;
#pragma unroll
                    for (int p = 0; p < DIM; ++p) ck[p] += 4 * lo * xl[(DEG == 1 ? 0 : NV + edge_between<DIM>(k, o))][p];
                }
        }
#pragma unroll
        for (int p = 0; p < DIM; ++p)
#pragma unroll
            for (int r = 0; r < DIM; ++r) G[p][r] += ck[p] * gl[k][r];
    }
}

template <int DIM, int DEG>
DEV void sym_grad_u_at(const double (&xl)[(DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6)][DIM], const double (&gl)[DIM + 1][DIM],
                       int q, double (&ef)[DIM * (DIM + 1) / 2]) {
    double G[DIM][DIM];
    grad_u_at<DIM, DEG>(xl, gl, q, G);
#pragma unroll
    for (int p = 0; p < DIM; ++p)
#pragma unroll
        for (int r = p; r < DIM; ++r) ef[flat_idx<DIM>(p, r)] = 0.5 * (G[p][r] + G[r][p]);
}

// Strain (or stress) field as per-element interpolants (Simulator::strainField / stressField, LinearElasticity.hh:511-526;
// Element::strain :99-117): the nodal values of the degree-(DEG-1) interpolant, i.e. one value per element for P1 and the
// values at the NV corners for P2. out: [nElem][NQ][flatLen].
template <int DIM, int DEG, int MAT>
__global__ void __launch_bounds__(256) k_strain_field(LoadArgs a, const double *__restrict__ uNodes, int wantStress, double *__restrict__ out) {
    constexpr int NV = DIM + 1;
    constexpr int FL = DIM * (DIM + 1) / 2;
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    constexpr int NQ = DEG == 1 ? 1 : NV;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.nElem; e += (int64_t)gridDim.x * 256) {
        const double *g = a.geo + e * a.geoStride;
        const int32_t *en = a.elemNodes + e * NPE;
        double gl[NV][DIM], xl[NPE][DIM];
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int d = 0; d < DIM; ++d) gl[k][d] = g[k * DIM + d];
#pragma unroll
        for (int j = 0; j < NPE; ++j) {
            const int64_t node = en[j];
#pragma unroll
            for (int d = 0; d < DIM; ++d) xl[j][d] = uNodes[node * DIM + d];
        }
        for (int q = 0; q < NQ; ++q) {
            double G[DIM][DIM], ef[FL];
            grad_u_at<DIM, DEG, true>(xl, gl, q, G);
#pragma unroll
            for (int p = 0; p < DIM; ++p)
#pragma unroll
                for (int r = p; r < DIM; ++r) ef[flat_idx<DIM>(p, r)] = 0.5 * (G[p][r] + G[r][p]);
            if (wantStress) {
                double sd[FL], sg[FL];
#pragma unroll
                for (int c = 0; c < FL; ++c) sd[c] = ef[c] * (c < DIM ? 1.0 : 2.0);
                elem_D_apply<DIM, MAT>(g, sd, sg);
#pragma unroll
                for (int c = 0; c < FL; ++c) ef[c] = sg[c];
            }
#pragma unroll
            for (int c = 0; c < FL; ++c) out[(e * NQ + q) * FL + c] = ef[c];
        }
    }
}

// The exact differential of the mutual energies with respect to every vertex coordinate
// (homogenizedElasticityTensorDiscreteDifferential, PeriodicHomogenization.hh:372-480, before the division by |Y|):
// for the unit perturbation e_c of vertex v, delta vol / vol = gl_v[c] and delta grad lambda_i = -gl_v gl_i[c], so that
// (delta eps)(w) = -sym(grad w[:, c] (x) gl_v) and the element's contribution to d/dp_v is  Q gl_v  with the
// Eshelby-like tensor
//     Q = int (G^ij : S^kl) I - (grad w^ij)^T S^kl - (grad w^kl)^T S^ij dV,     S = C : G, G = e + eps(w).
// One workgroup column per tensor entry ij <= kl (blockIdx.y); out: [pair][nVert][DIM], atomics per element corner.
template <int DIM, int DEG, int MAT>
__global__ void __launch_bounds__(256) k_mutual_energy_differential(LoadArgs a, const double *__restrict__ w, int64_t nNode,
                                                                    int64_t nVert, double *__restrict__ out) {
    constexpr int NV = DIM + 1;
    constexpr int FL = DIM * (DIM + 1) / 2;
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    constexpr int NQ = DEG == 1 ? 1 : NV;
    int ij = 0, rem = blockIdx.y;
    while (rem >= FL - ij) { rem -= FL - ij; ++ij; }
    const int kl = ij + rem;
    const double *wij = w + (int64_t)ij * nNode * DIM, *wkl = w + (int64_t)kl * nNode * DIM;
    double *o = out + (int64_t)blockIdx.y * nVert * DIM;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.nElem; e += (int64_t)gridDim.x * 256) {
        const double *g = a.geo + e * a.geoStride;
        const int32_t *en = a.elemNodes + e * NPE;
        double gl[NV][DIM];
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int d = 0; d < DIM; ++d) gl[k][d] = g[k * DIM + d];
        double xa[NPE][DIM], xb[NPE][DIM];
#pragma unroll
        for (int j = 0; j < NPE; ++j) {
            const int64_t node = en[j];
#pragma unroll
            for (int d = 0; d < DIM; ++d) { xa[j][d] = wij[node * DIM + d]; xb[j][d] = wkl[node * DIM + d]; }
        }
        double Q[DIM][DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c)
#pragma unroll
            for (int r = 0; r < DIM; ++r) Q[c][r] = 0.0;
        for (int q = 0; q < NQ; ++q) {
            double Wa[DIM][DIM], Wb[DIM][DIM], Ga[FL], Gb[FL], Sa[FL], Sb[FL], sd[FL];
            grad_u_at<DIM, DEG>(xa, gl, q, Wa);
            grad_u_at<DIM, DEG>(xb, gl, q, Wb);
#pragma unroll
            for (int p = 0; p < DIM; ++p)
#pragma unroll
                for (int r = p; r < DIM; ++r) {
                    const int c = flat_idx<DIM>(p, r);
                    Ga[c] = 0.5 * (Wa[p][r] + Wa[r][p]) + ((c == ij) ? (c < DIM ? 1.0 : 0.5) : 0.0);
                    Gb[c] = 0.5 * (Wb[p][r] + Wb[r][p]) + ((c == kl) ? (c < DIM ? 1.0 : 0.5) : 0.0);
                }
#pragma unroll
            for (int c = 0; c < FL; ++c) sd[c] = Ga[c] * (c < DIM ? 1.0 : 2.0);
            elem_D_apply<DIM, MAT>(g, sd, Sa);
#pragma unroll
            for (int c = 0; c < FL; ++c) sd[c] = Gb[c] * (c < DIM ? 1.0 : 2.0);
            elem_D_apply<DIM, MAT>(g, sd, Sb);
            double E = 0.0;
#pragma unroll
            for (int c = 0; c < FL; ++c) E += (c < DIM ? 1.0 : 2.0) * Ga[c] * Sb[c];
#pragma unroll
            for (int c = 0; c < DIM; ++c)
#pragma unroll
                for (int r = 0; r < DIM; ++r) {
                    double t = 0.0;
#pragma unroll
                    for (int p = 0; p < DIM; ++p) t += Wa[p][c] * Sb[flat_idx<DIM>(p, r)] + Wb[p][c] * Sa[flat_idx<DIM>(p, r)];
                    Q[c][r] += ((c == r) ? E : 0.0) - t;
                }
        }
        const double sc = g[12] / NQ;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int64_t v = en[k];
#pragma unroll
            for (int c = 0; c < DIM; ++c) {
                double val = 0.0;
#pragma unroll
                for (int r = 0; r < DIM; ++r) val += Q[c][r] * gl[k][r];
                unsafeAtomicAdd(&o[v * DIM + c], sc * val);
            }
        }
    }
}

// Mutual energies of the cell-problem fields, sum_e int (e^ij + eps(w^ij)) : C : (e^kl + eps(w^kl)) dV for every pair
// ij <= kl (blockIdx.y), i.e. |Y| Ch in the energy form (PeriodicHomogenization.hh:146-186 without the averaging
// shortcut), and with deltaP their discrete shape derivative in the volume form quoted at
// PeriodicHomogenization.hh:484-491:  int rel G^ij:C:G^kl + (delta eps)(w^ij):C:G^kl + G^ij:C:(delta eps)(w^kl) dV
// (the terms in delta w vanish by the cell problems' stationarity). w: [flatLen][nNode][DIM] per-node fields.
template <int DIM, int DEG, int MAT>
__global__ void __launch_bounds__(256) k_mutual_energies(LoadArgs a, const double *__restrict__ w, int64_t nNode,
                                                         const double *__restrict__ deltaP, double *__restrict__ out) {
    constexpr int NV = DIM + 1;
    constexpr int FL = DIM * (DIM + 1) / 2;
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    constexpr int NQ = DEG == 1 ? 1 : NV;
    // pair index -> (ij, kl), ij <= kl, row-major over the upper triangle
    int ij = 0, rem = blockIdx.y;
    while (rem >= FL - ij) { rem -= FL - ij; ++ij; }
    const int kl = ij + rem;
    const double *wij = w + (int64_t)ij * nNode * DIM, *wkl = w + (int64_t)kl * nNode * DIM;
    double acc = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.nElem; e += (int64_t)gridDim.x * 256) {
        const double *g = a.geo + e * a.geoStride;
        const int32_t *en = a.elemNodes + e * NPE;
        double gl[NV][DIM], dgl[NV][DIM], rel = 0.0;
        if (deltaP) load_corner_perturbation<DIM>(g, en, deltaP, gl, dgl, rel);
        else {
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int d = 0; d < DIM; ++d) gl[k][d] = g[k * DIM + d];
        }
        double xa[NPE][DIM], xb[NPE][DIM];
#pragma unroll
        for (int j = 0; j < NPE; ++j) {
            const int64_t node = en[j];
#pragma unroll
            for (int d = 0; d < DIM; ++d) { xa[j][d] = wij[node * DIM + d]; xb[j][d] = wkl[node * DIM + d]; }
        }
        double sum = 0.0;
        for (int q = 0; q < NQ; ++q) {
            double Ga[FL], Gb[FL], Sa[FL], Sb[FL], sd[FL];
            sym_grad_u_at<DIM, DEG>(xa, gl, q, Ga);
            sym_grad_u_at<DIM, DEG>(xb, gl, q, Gb);
#pragma unroll
            for (int c = 0; c < FL; ++c) {   // + canonical basis strains (SymmetricMatrix.hh:405-413)
                Ga[c] += (c == ij) ? (c < DIM ? 1.0 : 0.5) : 0.0;
                Gb[c] += (c == kl) ? (c < DIM ? 1.0 : 0.5) : 0.0;
            }
#pragma unroll
            for (int c = 0; c < FL; ++c) sd[c] = Gb[c] * (c < DIM ? 1.0 : 2.0);
            elem_D_apply<DIM, MAT>(g, sd, Sb);
            if (deltaP) {
                double dA[FL], dB[FL];
                sym_grad_u_at<DIM, DEG>(xa, dgl, q, dA);
                sym_grad_u_at<DIM, DEG>(xb, dgl, q, dB);
#pragma unroll
                for (int c = 0; c < FL; ++c) sd[c] = Ga[c] * (c < DIM ? 1.0 : 2.0);
                elem_D_apply<DIM, MAT>(g, sd, Sa);
#pragma unroll
                for (int c = 0; c < FL; ++c) {
                    const double m = c < DIM ? 1.0 : 2.0;
                    sum += m * ((rel * Ga[c] + dA[c]) * Sb[c] + dB[c] * Sa[c]);
                }
            } else {
#pragma unroll
                for (int c = 0; c < FL; ++c) sum += (c < DIM ? 1.0 : 2.0) * Ga[c] * Sb[c];
            }
        }
        acc += sum * g[12] / NQ;
    }
    // wave reduction, one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(&out[blockIdx.y], acc);
}

// (delta K) u, Simulator::applyDeltaStiffnessMatrix (LinearElasticity.hh:1301-1328) with deltaPerElementStiffness
// (:306-330) never formed: delta f = vol [B(rel gl + dgl, gl) + B(gl, dgl)] u with B the bilinear form above
// (rel = delta vol / vol). u is a per-NODE field, the result a per-DoF field, exactly as in the reference.
template <int DIM, int DEG, int MAT>
__global__ void __launch_bounds__(256) k_apply_delta_K(LoadArgs a, const double *__restrict__ uNodes, const double *__restrict__ deltaP,
                                                       double *__restrict__ out) {
    constexpr int NV = DIM + 1;
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < a.nElem; e += (int64_t)gridDim.x * 256) {
        const double *g = a.geo + e * a.geoStride;
        const int32_t *en = a.elemNodes + e * NPE;
        double gl[NV][DIM], dgl[NV][DIM], glA[NV][DIM], rel;
        load_corner_perturbation<DIM>(g, en, deltaP, gl, dgl, rel);
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int d = 0; d < DIM; ++d) glA[k][d] = rel * gl[k][d] + dgl[k][d];
        double xl[NPE][DIM];
        int64_t dof[NPE];
#pragma unroll
        for (int j = 0; j < NPE; ++j) {
            const int64_t node = en[j];
            dof[j] = a.dofForNode ? (int64_t)a.dofForNode[node] : node;
#pragma unroll
            for (int d = 0; d < DIM; ++d) xl[j][d] = uNodes[node * DIM + d];
        }
        // first term kept in registers so that every (node, component) costs one atomic, not two
        double f1[NPE][DIM];
        elem_forces_bilinear<DIM, DEG, MAT>(g, g[12], glA, gl, xl, [&](int j, const double *fv) {
#pragma unroll
            for (int d = 0; d < DIM; ++d) f1[j][d] = fv[d];
        });
        elem_forces_bilinear<DIM, DEG, MAT>(g, g[12], gl, dgl, xl, [&](int j, const double *fv) {
#pragma unroll
            for (int d = 0; d < DIM; ++d) unsafeAtomicAdd(&out[dof[j] * DIM + d], f1[j][d] + fv[d]);
        });
    }
}

template <int DIM, int DEG, int MAT>
__global__ void __launch_bounds__(256) k_mf_forces(SpmvMfArgs a, const double *__restrict__ x, double *__restrict__ fbuf,
                                                   const double *scal, int it, const double *stopPtr) {
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    if (scal) {
        it += (int)stopPtr[3];
        if (scal[(int64_t)it * 4 + 2] <= stopPtr[0]) return;
    }
    const int64_t nE = a.nElem;
    // element groups of 256: XCD-contiguous ranges of groups, see xcd_span
    int64_t grpFirst = blockIdx.x, grpEnd = (nE + 255) / 256, grpStride = gridDim.x;
    if (a.xcd) xcd_span(grpEnd, grpFirst, grpEnd, grpStride);
    for (int64_t grp = grpFirst; grp < grpEnd; grp += grpStride) {
        const int64_t e = grp * 256 + threadIdx.x;
        if (e >= nE) continue;
        double f[NPE][DIM];
        elem_forces<DIM, DEG, MAT>(a, e, x, f);
        if (a.pairPos) {
            // list order: the pair's force goes where the rows pass will stream it
#pragma unroll
            for (int j = 0; j < NPE; ++j) {
                const uint32_t pp = a.pairPos[e * NPE + j];
                if (pp == 0xffffffffu) continue;
#pragma unroll
                for (int d = 0; d < DIM; ++d) fbuf[(int64_t)pp * DIM + d] = f[j][d];
            }
            continue;
        }
        // element-major AoS: the (element, node) pair with code e*NPE + i owns fbuf[code*DIM .. +DIM)
        double *o = fbuf + e * (NPE * DIM);
        if ((NPE * DIM) % 2 == 0) {
            typedef double dv2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int q = 0; q < NPE * DIM / 2; ++q) {
                dv2 w = {f[(2 * q) / DIM][(2 * q) % DIM], f[(2 * q + 1) / DIM][(2 * q + 1) % DIM]};
                *reinterpret_cast<dv2 *>(o + 2 * q) = w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < NPE * DIM; ++q) o[q] = f[q / DIM][q % DIM];
        }
    }
}

// Cluster variant: one workgroup per block of MF_BLOCK consecutive elements. The nodal forces of the block are summed
// in LDS (one accumulator per distinct row of the block). Rows whose elements all belong to the block are finished
// here and written to y; the other rows leave ONE partial sum per (block, row) in the interface buffer, in row order,
// which k_mf_rows sums. With a spatially coherent element order most rows are finished in their block, so the
// 240 B/element force buffer of k_mf_forces shrinks to a few tens of bytes per element.
template <int DIM, int DEG, int MAT, bool PCG>
__global__ void __launch_bounds__(MF_BLOCK) k_mf_cluster(SpmvMfArgs a, const double *__restrict__ x, double *__restrict__ y, double *dotOut,
                                                         double *scal, int it, const double *stopPtr) {
    constexpr int NPE = (DIM == 3) ? (DEG == 1 ? 4 : 10) : (DEG == 1 ? 3 : 6);
    extern __shared__ __attribute__((aligned(16))) double clacc[];   // accumulators [maxLocal * DIM] + staged x [maxLocal * DIM] + 16
    double *xs = clacc + a.clMaxLocal * DIM;
    double *red = xs + a.clMaxLocal * DIM;
    if (PCG) {
        it += (int)stopPtr[3];
        if (scal[(int64_t)it * 4 + 2] <= stopPtr[0]) return;
        dotOut = scal + (int64_t)it * 4 + 1;
    }
    double dot = 0.0;
    for (int64_t b = blockIdx.x; b < a.clBlocks; b += gridDim.x) {
        const int u0 = a.clBlockPtr[b], nLocal = a.clBlockPtr[b + 1] - u0;
        // x of every distinct row of the block is read ONCE into LDS (a block of 256 P2 tets gathers 2560 nodal vectors
        // but touches only ~580 distinct rows); the lanes then pick their 10 vectors from LDS by local row index
        for (int t = threadIdx.x; t < nLocal; t += MF_BLOCK) {
            const int64_t row = a.clEntryRow[u0 + t];
#pragma unroll
            for (int d = 0; d < DIM; ++d) { xs[t * DIM + d] = x[row * DIM + d]; clacc[t * DIM + d] = 0.0; }
        }
        __syncthreads();
        const int64_t e = b * a.clBlockElems + threadIdx.x;
        if ((int)threadIdx.x < a.clBlockElems && e < a.nElem) {
            int li[NPE];
            double xl[NPE][DIM];
#pragma unroll
            for (int j = 0; j < NPE; ++j) {
                li[j] = a.clLocalIdx[e * NPE + j];
#pragma unroll
                for (int d = 0; d < DIM; ++d) xl[j][d] = xs[li[j] * DIM + d];
            }
            elem_forces_core<DIM, DEG, MAT>(a, e, xl, [&](int j, const double *fv) {
#pragma unroll
                for (int d = 0; d < DIM; ++d) unsafeAtomicAdd(&clacc[li[j] * DIM + d], fv[d]);
            });
        }
        __syncthreads();
        for (int t = threadIdx.x; t < nLocal; t += MF_BLOCK) {
            const int dest = a.clEntryDest[u0 + t];
            if (dest == -2) continue;                                  // row owned by another rank
            if (dest >= 0) {
#pragma unroll
                for (int d = 0; d < DIM; ++d) a.clIfaceBuf[(int64_t)dest * DIM + d] = clacc[t * DIM + d];
                continue;
            }
            const int64_t row = a.clEntryRow[u0 + t];
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                const int64_t gi = row * DIM + d;
                double v = clacc[t * DIM + d];
                if (a.fixedMask && a.fixedMask[gi]) v = 0.0;
                y[gi] = v;
                if (dotOut) dot += v * x[gi];
            }
        }
        __syncthreads();
    }
    if (dotOut) {
        double v[1] = {dot};
        block_sum<1>(v, red);
        if (threadIdx.x == 0) unsafeAtomicAdd(dotOut, v[0]);
    }
}

// y_row = sum over the (element, node) pairs of the row of their nodal force: a pure gather-sum
template <int DIM, bool PCG>
__global__ void __launch_bounds__(256) k_mf_rows(SpmvMfArgs a, const double *__restrict__ fbuf, const double *__restrict__ x,
                                                 double *__restrict__ y, double *dotOut, double *scal, int it, const double *stopPtr) {
    extern __shared__ __attribute__((aligned(16))) double mfacc[];   // [maxRows * DIM] + 16
    double *red = mfacc + a.maxRows * DIM;
    if (PCG) {
        it += (int)stopPtr[3];
        if (scal[(int64_t)it * 4 + 2] <= stopPtr[0]) return;
        dotOut = scal + (int64_t)it * 4 + 1;
    }
    double dot = 0.0;
    int64_t chunkFirst = blockIdx.x, chunkEnd = a.nChunk, chunkStride = gridDim.x;
    if (a.xcd) xcd_span(a.nChunk, chunkFirst, chunkEnd, chunkStride);
    for (int64_t chunk = chunkFirst; chunk < chunkEnd; chunk += chunkStride) {
        const int r0 = a.chunkRow[chunk], r1 = a.chunkRow[chunk + 1];
        const int nr = (r1 - r0) * DIM;
        for (int t = threadIdx.x; t < nr; t += 256) mfacc[t] = 0.0;
        __syncthreads();
        const int64_t kb = a.pairPtr[chunk], ke = a.pairPtr[chunk + 1];
        // U independent pairs per lane and trip: their list and force loads are all in flight together
        constexpr int U = 2;
        for (int64_t k0 = kb + threadIdx.x; k0 < ke; k0 += 256 * U) {
            int64_t code[U];
            int lr[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t kk = k0 + (int64_t)u * 256;
                ok[u] = kk < ke;
                code[u] = a.pairPos ? (ok[u] ? kk : kb) : (int64_t)(ok[u] ? a.pairCode[kk] : a.pairCode[kb]);
                lr[u] = ok[u] ? (int)a.pairRow[kk] : 0;
            }
            double out[U][DIM];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < DIM; ++c) out[u][c] = fbuf[code[u] * DIM + c];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
#pragma unroll
                for (int c = 0; c < DIM; ++c) unsafeAtomicAdd(&mfacc[lr[u] * DIM + c], out[u][c]);
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < nr; idx += 256) {
            int64_t gi = (int64_t)r0 * DIM + idx;
            if (a.rowMap) gi = (int64_t)a.rowMap[r0 + idx / DIM] * DIM + idx % DIM;   // cluster variant: compact numbering of the interface rows
            double v = mfacc[idx];
            if (a.fixedMask && a.fixedMask[gi]) v = 0.0;
            y[gi] = v;
            if (dotOut) dot += v * x[gi];
        }
        __syncthreads();
    }
    if (dotOut) {
        double v[1] = {dot};
        block_sum<1>(v, red);
        if (threadIdx.x == 0) unsafeAtomicAdd(dotOut, v[0]);
    }
}

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
        const int mid = (lo + hi) >> 1;
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

// rc[a*nModes + k] = sum_{DoFs n of aggregate a} z_{a,k}(n) . w(n); one workgroup per aggregate
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
                                                     const double *__restrict__ vals, double *__restrict__ Ac) {
    constexpr int NB = DIM * DIM;
    constexpr int NM = DIM == 3 ? 6 : 3;
    constexpr int NSLOT = DIM == 3 ? 27 : 9;
    __shared__ double nbr[NSLOT * NM * NM];
    __shared__ int nbrAgg[NSLOT];
    __shared__ double diagRed[16 * NM * NM];
    const int a = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    const int64_t m = (int64_t)t.nAgg * NM;
    for (int q = threadIdx.x; q < NSLOT * NM * NM; q += blockDim.x) nbr[q] = 0.0;
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
                continue;
            }
            int slot = -1;
            if (binCoord) {
                const int dx = binCoord[b * 3] - ca[0], dy = binCoord[b * 3 + 1] - ca[1], dz = binCoord[b * 3 + 2] - ca[2];
                if (dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1) slot = (dx + 1) + 3 * (dy + 1) + (DIM == 3 ? 9 * (dz + 1) : 0);
            }
            if (slot >= 0) {
                nbrAgg[slot] = b;     // every writer stores the same value
#pragma unroll
                for (int q = 0; q < NM * NM; ++q) unsafeAtomicAdd(&nbr[slot * NM * NM + q], T[q]);
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
                                                  const double *stopPtr) {
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
        if (threadIdx.x == 0) unsafeAtomicAdd(rzOut, acc[0]);
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
                                                  double *__restrict__ p, double *scal) {
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
    if (threadIdx.x == 0) { unsafeAtomicAdd(&scal[0], acc[0]); unsafeAtomicAdd(&scal[2], acc[1]); }
}

// x += alpha p ; r -= alpha Ap ; z = Dinv r ; scal[it+1].{rz,rr} += ...
template <int DIM, bool SKIPZ = false>
__global__ void __launch_bounds__(256) k_pcg_update(int64_t nRows, const double *__restrict__ dinv, const double *__restrict__ p,
                                                    const double *__restrict__ Ap, double *__restrict__ x, double *__restrict__ r,
                                                    double *__restrict__ z, double *scal, int it, const double *stopPtr) {
    __shared__ double red[16];
    it += (int)stopPtr[3];
    if (scal[(int64_t)it * 4 + 2] <= stopPtr[0]) return;
    const double alpha = scal[(int64_t)it * 4 + 0] / scal[(int64_t)it * 4 + 1];
    double acc[2] = {0, 0};
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < nRows; n += (int64_t)gridDim.x * 256) {
        double rv[DIM], zv[DIM];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
            const int64_t g = n * DIM + c;
            x[g] += alpha * p[g];
            rv[c] = r[g] - alpha * Ap[g];
            r[g] = rv[c];
        }
        if (!SKIPZ) {
            apply_block<DIM>(dinv + n * (DIM * (DIM + 1) / 2), rv, zv);
#pragma unroll
            for (int c = 0; c < DIM; ++c) { z[n * DIM + c] = zv[c]; acc[0] += rv[c] * zv[c]; }
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) acc[1] += rv[c] * rv[c];
    }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) {
        if (!SKIPZ) unsafeAtomicAdd(&scal[(int64_t)(it + 1) * 4 + 0], acc[0]);
        unsafeAtomicAdd(&scal[(int64_t)(it + 1) * 4 + 2], acc[1]);
    }
}

// p = z + beta p
__global__ void __launch_bounds__(256) k_pcg_direction(int64_t n, const double *__restrict__ z, double *__restrict__ p,
                                                       const double *scal, int it, const double *stopPtr) {
    it += (int)stopPtr[3];
    if (scal[(int64_t)it * 4 + 2] <= stopPtr[0]) return;
    const double beta = scal[(int64_t)(it + 1) * 4 + 0] / scal[(int64_t)it * 4 + 0];
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) p[k] = z[k] + beta * p[k];
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
__global__ void __launch_bounds__(256) k_dev_dots(int64_t n, const double *__restrict__ r, const double *__restrict__ z, double *out) {
    __shared__ double red[16];
    double acc[2] = {0, 0};
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) { acc[0] += r[k] * z[k]; acc[1] += r[k] * r[k]; }
    block_sum<2>(acc, red);
    if (threadIdx.x == 0) { unsafeAtomicAdd(&out[0], acc[0]); unsafeAtomicAdd(&out[1], acc[1]); }
}

// stop[3] += n: advances the iteration base at the end of a captured block of PCG iterations
__global__ void k_advance_base(double *stop, double n) { stop[3] += n; }

__global__ void __launch_bounds__(256) k_axpby(int64_t n, double a, const double *__restrict__ x, double b, double *__restrict__ y) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        y[k] = a * x[k] + (b == 0.0 ? 0.0 : b * y[k]);
}
__global__ void __launch_bounds__(256) k_mask(int64_t n, const uint8_t *__restrict__ m, double *__restrict__ v) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256)
        if (m[k]) v[k] = 0.0;
}
__global__ void __launch_bounds__(256) k_scatter_values(int64_t n, const int64_t *__restrict__ idx, const double *__restrict__ val,
                                                        double *__restrict__ v) {
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) v[idx[k]] = val[k];
}
__global__ void __launch_bounds__(256) k_dot(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *out) {
    __shared__ double red[8];
    double acc[1] = {0};
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) acc[0] += a[k] * b[k];
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(out, acc[0]);
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
// launchers
// ------------------------------------------------------------------------------------------------
static inline int grid_for(int64_t n, int cap = 2048) {
    int64_t g = (n + 255) / 256;
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}
#define CHECK_LAUNCH() MFH_HIP(hipGetLastError())

void launch_geometry(int dim, int /*deg*/, int /*mat*/, int64_t nElem, const int32_t *elemNodes, int npe, const double *vertPos,
                     const double *matParams, int matMode, double *geo, int geoStride, int *negCount, hipStream_t s) {
    const int grid = (int)((nElem + 255) / 256);
    if (dim == 3) hipLaunchKernelGGL(k_geometry<3>, dim3(grid), dim3(256), 0, s, nElem, elemNodes, npe, vertPos, matParams, matMode, geo, geoStride, negCount);
    else hipLaunchKernelGGL(k_geometry<2>, dim3(grid), dim3(256), 0, s, nElem, elemNodes, npe, vertPos, matParams, matMode, geo, geoStride, negCount);
    CHECK_LAUNCH();
}

// dispatch on (dim, deg, mat)
#define MFH_DISPATCH(a, CALL)                                                            \
    do {                                                                                 \
        const int key_ = (a.dim == 3 ? 0 : 4) + (a.deg == 2 ? 2 : 0) + (a.mat == MAT_GENERAL ? 1 : 0); \
        switch (key_) {                                                                  \
        case 0: { CALL(3, 1, MAT_ISO); } break;                                           \
        case 1: { CALL(3, 1, MAT_GENERAL); } break;                                       \
        case 2: { CALL(3, 2, MAT_ISO); } break;                                           \
        case 3: { CALL(3, 2, MAT_GENERAL); } break;                                       \
        case 4: { CALL(2, 1, MAT_ISO); } break;                                           \
        case 5: { CALL(2, 1, MAT_GENERAL); } break;                                       \
        case 6: { CALL(2, 2, MAT_ISO); } break;                                           \
        default: { CALL(2, 2, MAT_GENERAL); } break;                                      \
        }                                                                                \
    } while (0)

// assembly kernels also come in the scalar-operator flavours
#define MFH_DISPATCH_ASM(a, CALL)                                                        \
    do {                                                                                 \
        if (!mat_is_scalar(a.mat)) { MFH_DISPATCH(a, CALL); break; }                     \
        const int key_ = (a.dim == 3 ? 0 : 4) + (a.deg == 2 ? 2 : 0) + (a.mat == MAT_MASS ? 1 : 0); \
        switch (key_) {                                                                  \
        case 0: { CALL(3, 1, MAT_LAPLACE); } break;                                       \
        case 1: { CALL(3, 1, MAT_MASS); } break;                                          \
        case 2: { CALL(3, 2, MAT_LAPLACE); } break;                                       \
        case 3: { CALL(3, 2, MAT_MASS); } break;                                          \
        case 4: { CALL(2, 1, MAT_LAPLACE); } break;                                       \
        case 5: { CALL(2, 1, MAT_MASS); } break;                                          \
        case 6: { CALL(2, 2, MAT_LAPLACE); } break;                                       \
        default: { CALL(2, 2, MAT_MASS); } break;                                         \
        }                                                                                \
    } while (0)

void launch_assemble_gather(const AsmArgs &a, hipStream_t s) {
    if (a.nChunk == 0) return;
    const size_t lds = (size_t)(mat_is_scalar(a.mat) ? 1 : a.dim * a.dim) * (a.chunkSlots + 2) * sizeof(double);
#define CALL(D, G, M)                                                                                          \
    if (lds > 64 * 1024)                                                                                         \
        MFH_HIP(hipFuncSetAttribute((const void *)k_assemble_gather<D, G, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_assemble_gather<D, G, M>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a)
    if (a.debugVariant && a.dim == 3 && a.deg == 2 && a.mat == MAT_ISO) {
        switch (a.debugVariant) {
        case 1: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 1>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 2>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 21: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 21>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 22: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 22>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 23: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 23>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 24: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 24>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 28: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 28>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 25: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 25>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 26: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 26>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 27: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 27>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 11: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 11>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 12: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 12>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 13: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 13>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 14: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 14>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 16: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 16>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        case 18: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 18>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        default: hipLaunchKernelGGL((k_assemble_gather<3, 2, MAT_ISO, 0>), dim3((unsigned)a.nChunk), dim3(256), lds, s, a); break;
        }
    } else {
        MFH_DISPATCH_ASM(a, CALL);
    }
#undef CALL
    CHECK_LAUNCH();
}

void launch_assemble_atomic(const AsmArgs &a, hipStream_t s) {
    const int64_t total = a.nElem * a.npe * a.npe;
    const int grid = grid_for(total, 256 * 32);
#define CALL(D, G, M) hipLaunchKernelGGL((k_assemble_atomic<D, G, M>), dim3(grid), dim3(256), 0, s, a)
    MFH_DISPATCH_ASM(a, CALL);
#undef CALL
    CHECK_LAUNCH();
}

void launch_element_stiffness(const AsmArgs &a, int64_t first, int64_t count, double *KeOut, hipStream_t s) {
    const int grid = grid_for(count * a.npe * a.npe);
#define CALL(D, G, M) hipLaunchKernelGGL((k_element_stiffness<D, G, M>), dim3(grid), dim3(256), 0, s, a, first, count, KeOut)
    MFH_DISPATCH_ASM(a, CALL);
#undef CALL
    CHECK_LAUNCH();
}

static LoadArgs make_load_args(const AsmArgs &a, const int32_t *elemNodes, const int32_t *dofForNode, const double *intGrad,
                               const double *cstrain) {
    LoadArgs l{};
    l.nElem = a.nElem; l.npe = a.npe; l.geoStride = a.geoStride; l.geo = a.geo; l.elemNodes = elemNodes; l.dofForNode = dofForNode;
    for (int k = 0; k < 2 * a.npe; ++k) l.intGrad[k] = intGrad[k];
    for (int k = 0; k < 6; ++k) l.cstrain[k] = cstrain ? cstrain[k] : 0.0;
    return l;
}

void launch_constant_strain_load(const AsmArgs &a, const int32_t *elemNodes, const int32_t *dofForNode, const double *intGrad,
                                 const double *cstrain, const double *deltaP, double *out, hipStream_t s) {
    const LoadArgs l = make_load_args(a, elemNodes, dofForNode, intGrad, cstrain);
    const int grid = grid_for(a.nElem * a.npe, 8192);
#define CALL(D, G, M) hipLaunchKernelGGL((k_constant_strain_load<D, G, M>), dim3(grid), dim3(256), 0, s, l, deltaP, out)
    MFH_DISPATCH(a, CALL);
#undef CALL
    CHECK_LAUNCH();
}

void launch_average_strain(const AsmArgs &a, const int32_t *elemNodes, const double *intGrad, const double *uNodes, double *out,
                           int wantStress, const double *uFixed, const double *deltaP, hipStream_t s) {
    const LoadArgs l = make_load_args(a, elemNodes, nullptr, intGrad, nullptr);
    const int grid = grid_for(a.nElem, 8192);
#define CALL(D, G, M) hipLaunchKernelGGL((k_average_strain<D, G, M>), dim3(grid), dim3(256), 0, s, l, uNodes, out, wantStress, uFixed, deltaP)
    MFH_DISPATCH(a, CALL);
#undef CALL
    CHECK_LAUNCH();
}

void launch_apply_delta_K(const AsmArgs &a, const int32_t *elemNodes, const int32_t *dofForNode, const double *intGrad,
                          const double *uNodes, const double *deltaP, double *out, hipStream_t s) {
    const LoadArgs l = make_load_args(a, elemNodes, dofForNode, intGrad, nullptr);
    const int grid = grid_for(a.nElem, 8192);
#define CALL(D, G, M) hipLaunchKernelGGL((k_apply_delta_K<D, G, M>), dim3(grid), dim3(256), 0, s, l, uNodes, deltaP, out)
    MFH_DISPATCH(a, CALL);
#undef CALL
    CHECK_LAUNCH();
}

void launch_mutual_energies(const AsmArgs &a, const int32_t *elemNodes, const double *intGrad, const double *w, int64_t nNode,
                            const double *deltaP, double *out, hipStream_t s) {
    const LoadArgs l = make_load_args(a, elemNodes, nullptr, intGrad, nullptr);
    const int fl = a.dim * (a.dim + 1) / 2;
    const dim3 grid(grid_for(a.nElem, 4096), fl * (fl + 1) / 2);
#define CALL(D, G, M) hipLaunchKernelGGL((k_mutual_energies<D, G, M>), grid, dim3(256), 0, s, l, w, nNode, deltaP, out)
    MFH_DISPATCH(a, CALL);
#undef CALL
    CHECK_LAUNCH();
}

void launch_strain_field(const AsmArgs &a, const int32_t *elemNodes, const double *intGrad, const double *uNodes, int wantStress,
                         double *out, hipStream_t s) {
    const LoadArgs l = make_load_args(a, elemNodes, nullptr, intGrad, nullptr);
    const int grid = grid_for(a.nElem, 8192);
#define CALL(D, G, M) hipLaunchKernelGGL((k_strain_field<D, G, M>), dim3(grid), dim3(256), 0, s, l, uNodes, wantStress, out)
    MFH_DISPATCH(a, CALL);
#undef CALL
    CHECK_LAUNCH();
}

void launch_mutual_energy_differential(const AsmArgs &a, const int32_t *elemNodes, const double *intGrad, const double *w,
                                       int64_t nNode, int64_t nVert, double *out, hipStream_t s) {
    const LoadArgs l = make_load_args(a, elemNodes, nullptr, intGrad, nullptr);
    const int fl = a.dim * (a.dim + 1) / 2;
    const dim3 grid(grid_for(a.nElem, 4096), fl * (fl + 1) / 2);
#define CALL(D, G, M) hipLaunchKernelGGL((k_mutual_energy_differential<D, G, M>), grid, dim3(256), 0, s, l, w, nNode, nVert, out)
    MFH_DISPATCH(a, CALL);
#undef CALL
    CHECK_LAUNCH();
}

void launch_average_gradient(const AsmArgs &a, const int32_t *elemNodes, const double *intGrad, const double *uNodes, double *out,
                             hipStream_t s) {
    const LoadArgs l = make_load_args(a, elemNodes, nullptr, intGrad, nullptr);
    const int grid = grid_for(a.nElem, 8192);
    if (a.dim == 3) {
        if (a.deg == 2) hipLaunchKernelGGL((k_average_gradient<3, 2>), dim3(grid), dim3(256), 0, s, l, uNodes, out);
        else hipLaunchKernelGGL((k_average_gradient<3, 1>), dim3(grid), dim3(256), 0, s, l, uNodes, out);
    } else {
        if (a.deg == 2) hipLaunchKernelGGL((k_average_gradient<2, 2>), dim3(grid), dim3(256), 0, s, l, uNodes, out);
        else hipLaunchKernelGGL((k_average_gradient<2, 1>), dim3(grid), dim3(256), 0, s, l, uNodes, out);
    }
    CHECK_LAUNCH();
}

// persistent workgroups; a multiple of 8 (xcd_span)
static int persistent_grid(int64_t nItems, int cap) { return (int)std::max<int64_t>(8, std::min<int64_t>(cap, (nItems + 7) / 8 * 8)); }
static int spmv_grid(const SpmvArgs &a) { return persistent_grid(a.nChunk, 256 * 8); }

void launch_spmv(const SpmvArgs &a, const double *x, double *y, double *dotOut, hipStream_t s) {
    if (a.nChunk == 0) return;
    const size_t lds = ((size_t)a.dim * a.chunkSlots + 16) * sizeof(double);
    if (a.dim == 1) hipLaunchKernelGGL((k_spmv<1, false>), dim3(spmv_grid(a)), dim3(256), lds, s, a, x, y, dotOut, (double *)nullptr, 0, (const double *)nullptr);
    else if (a.dim == 3) hipLaunchKernelGGL((k_spmv<3, false>), dim3(spmv_grid(a)), dim3(256), lds, s, a, x, y, dotOut, (double *)nullptr, 0, (const double *)nullptr);
    else hipLaunchKernelGGL((k_spmv<2, false>), dim3(spmv_grid(a)), dim3(256), lds, s, a, x, y, dotOut, (double *)nullptr, 0, (const double *)nullptr);
    CHECK_LAUNCH();
}

void launch_pcg_spmv(const SpmvArgs &a, const double *p, double *Ap, double *scal, int it, const double *stopPtr, hipStream_t s) {
    if (a.nChunk == 0) return;
    const size_t lds = ((size_t)a.dim * a.chunkSlots + 16) * sizeof(double);
    if (a.dim == 1) hipLaunchKernelGGL((k_spmv<1, true>), dim3(spmv_grid(a)), dim3(256), lds, s, a, p, Ap, (double *)nullptr, scal, it, stopPtr);
    else if (a.dim == 3) hipLaunchKernelGGL((k_spmv<3, true>), dim3(spmv_grid(a)), dim3(256), lds, s, a, p, Ap, (double *)nullptr, scal, it, stopPtr);
    else hipLaunchKernelGGL((k_spmv<2, true>), dim3(spmv_grid(a)), dim3(256), lds, s, a, p, Ap, (double *)nullptr, scal, it, stopPtr);
    CHECK_LAUNCH();
}

void launch_spmv_mf(const SpmvMfArgs &a, const double *x, double *y, double *dotOut, double *scal, int it, const double *stopPtr,
                    bool pcg, hipStream_t s) {
    if (a.nChunk == 0) return;
    const int bs = mat_is_scalar(a.mat) ? 1 : a.dim;
    const size_t lds = ((size_t)a.maxRows * bs + 16) * sizeof(double);
    const int grid = persistent_grid(a.nChunk, 256 * 8);
#define CALL(D, G, M)                                                                                                          \
    if (pcg) hipLaunchKernelGGL((k_spmv_mf<D, G, M, true>), dim3(grid), dim3(256), lds, s, a, x, y, (double *)nullptr, scal, it, stopPtr); \
    else hipLaunchKernelGGL((k_spmv_mf<D, G, M, false>), dim3(grid), dim3(256), lds, s, a, x, y, dotOut, (double *)nullptr, 0, (const double *)nullptr)
    if (a.variant && a.dim == 3 && a.deg == 2 && a.mat == MAT_ISO) {
        // timing experiments: unroll factor of the j loop (register pressure vs ILP)
        switch (a.variant) {
        case 1: if (pcg) hipLaunchKernelGGL((k_spmv_mf<3, 2, MAT_ISO, true, 1>), dim3(grid), dim3(256), lds, s, a, x, y, (double *)nullptr, scal, it, stopPtr);
                else hipLaunchKernelGGL((k_spmv_mf<3, 2, MAT_ISO, false, 1>), dim3(grid), dim3(256), lds, s, a, x, y, dotOut, (double *)nullptr, 0, (const double *)nullptr); break;
        case 2: if (pcg) hipLaunchKernelGGL((k_spmv_mf<3, 2, MAT_ISO, true, 2>), dim3(grid), dim3(256), lds, s, a, x, y, (double *)nullptr, scal, it, stopPtr);
                else hipLaunchKernelGGL((k_spmv_mf<3, 2, MAT_ISO, false, 2>), dim3(grid), dim3(256), lds, s, a, x, y, dotOut, (double *)nullptr, 0, (const double *)nullptr); break;
        default: if (pcg) hipLaunchKernelGGL((k_spmv_mf<3, 2, MAT_ISO, true, 5>), dim3(grid), dim3(256), lds, s, a, x, y, (double *)nullptr, scal, it, stopPtr);
                else hipLaunchKernelGGL((k_spmv_mf<3, 2, MAT_ISO, false, 5>), dim3(grid), dim3(256), lds, s, a, x, y, dotOut, (double *)nullptr, 0, (const double *)nullptr); break;
        }
    } else {
        MFH_DISPATCH_ASM(a, CALL);
    }
#undef CALL
    CHECK_LAUNCH();
}

// two-pass matrix-free elasticity operator (k_mf_stress + k_mf_rows)
void launch_spmv_mf2(const SpmvMfArgs &a, const double *x, double *y, double *dotOut, double *scal, int it, const double *stopPtr,
                     bool pcg, hipStream_t s) {
    if (a.nChunk == 0) return;
    const int gridE = persistent_grid((a.nElem + 255) / 256, 256 * 32);   // persistent over element groups
#define CALL(D, G, M) hipLaunchKernelGGL((k_mf_forces<D, G, M>), dim3(gridE), dim3(256), 0, s, a, x, a.sig, pcg ? (const double *)scal : (const double *)nullptr, it, stopPtr)
    MFH_DISPATCH(a, CALL);
#undef CALL
    CHECK_LAUNCH();
    const size_t lds = ((size_t)a.maxRows * a.dim + 16) * sizeof(double);
    const int grid = persistent_grid(a.nChunk, 256 * 8);
#define ROWS(D)                                                                                                                      \
    if (pcg) hipLaunchKernelGGL((k_mf_rows<D, true>), dim3(grid), dim3(256), lds, s, a, (const double *)a.sig, x, y, (double *)nullptr, scal, it, stopPtr); \
    else hipLaunchKernelGGL((k_mf_rows<D, false>), dim3(grid), dim3(256), lds, s, a, (const double *)a.sig, x, y, dotOut, (double *)nullptr, 0, (const double *)nullptr)
    if (a.dim == 3) { ROWS(3); } else { ROWS(2); }
#undef ROWS
    CHECK_LAUNCH();
}

// cluster variant of the matrix-free elasticity operator (k_mf_cluster + k_mf_rows over the interface partials)
void launch_spmv_mf_cluster(const SpmvMfArgs &a, const double *x, double *y, double *dotOut, double *scal, int it, const double *stopPtr,
                            bool pcg, hipStream_t s) {
    const size_t ldsC = ((size_t)2 * a.clMaxLocal * a.dim + 16) * sizeof(double);
    const int gridC = (int)std::min<int64_t>(a.clBlocks, 256 * 64);
#define CALL(D, G, M)                                                                                                                  \
    if (pcg) hipLaunchKernelGGL((k_mf_cluster<D, G, M, true>), dim3(gridC), dim3(MF_BLOCK), ldsC, s, a, x, y, (double *)nullptr, scal, it, stopPtr); \
    else hipLaunchKernelGGL((k_mf_cluster<D, G, M, false>), dim3(gridC), dim3(MF_BLOCK), ldsC, s, a, x, y, dotOut, (double *)nullptr, 0, (const double *)nullptr)
    MFH_DISPATCH(a, CALL);
#undef CALL
    CHECK_LAUNCH();
    if (a.nChunk == 0) return;
    const size_t lds = ((size_t)a.maxRows * a.dim + 16) * sizeof(double);
    const int grid = persistent_grid(a.nChunk, 256 * 8);
#define ROWS(D)                                                                                                                      \
    if (pcg) hipLaunchKernelGGL((k_mf_rows<D, true>), dim3(grid), dim3(256), lds, s, a, (const double *)a.sig, x, y, (double *)nullptr, scal, it, stopPtr); \
    else hipLaunchKernelGGL((k_mf_rows<D, false>), dim3(grid), dim3(256), lds, s, a, (const double *)a.sig, x, y, dotOut, (double *)nullptr, 0, (const double *)nullptr)
    if (a.dim == 3) { ROWS(3); } else { ROWS(2); }
#undef ROWS
    CHECK_LAUNCH();
}

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
    if (dim == 1) hipLaunchKernelGGL(k_pcg_init<1>, dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, b, x, r, z, p, scal);
    else if (dim == 3) hipLaunchKernelGGL(k_pcg_init<3>, dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, b, x, r, z, p, scal);
    else hipLaunchKernelGGL(k_pcg_init<2>, dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, b, x, r, z, p, scal);
    CHECK_LAUNCH();
}

void launch_pcg_update(int dim, int64_t nRows, const double *dinv, const double *p, const double *Ap, double *x, double *r,
                       double *z, double *scal, int it, const double *stopPtr, hipStream_t s) {
    if (dim == 1) hipLaunchKernelGGL((k_pcg_update<1, false>), dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, p, Ap, x, r, z, scal, it, stopPtr);
    else if (dim == 3) hipLaunchKernelGGL((k_pcg_update<3, false>), dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, p, Ap, x, r, z, scal, it, stopPtr);
    else hipLaunchKernelGGL((k_pcg_update<2, false>), dim3(grid_for(nRows)), dim3(256), 0, s, nRows, dinv, p, Ap, x, r, z, scal, it, stopPtr);
    CHECK_LAUNCH();
}

void launch_pcg_update_noz(int dim, int64_t nRows, const double *p, const double *Ap, double *x, double *r, double *scal, int it,
                           const double *stopPtr, hipStream_t s) {
    if (dim == 3) hipLaunchKernelGGL((k_pcg_update<3, true>), dim3(grid_for(nRows)), dim3(256), 0, s, nRows, (const double *)nullptr, p, Ap, x, r, (double *)nullptr, scal, it, stopPtr);
    else hipLaunchKernelGGL((k_pcg_update<2, true>), dim3(grid_for(nRows)), dim3(256), 0, s, nRows, (const double *)nullptr, p, Ap, x, r, (double *)nullptr, scal, it, stopPtr);
    CHECK_LAUNCH();
}

void launch_tl_fill(const TLArgs &t, const int32_t *colorOfAgg, int color, int mode, double *v, hipStream_t s) {
    if (t.dim == 3) hipLaunchKernelGGL(k_tl_fill<3>, dim3(grid_for(t.nDoF)), dim3(256), 0, s, t, colorOfAgg, color, mode, v);
    else hipLaunchKernelGGL(k_tl_fill<2>, dim3(grid_for(t.nDoF)), dim3(256), 0, s, t, colorOfAgg, color, mode, v);
    CHECK_LAUNCH();
}
void launch_tl_restrict(const TLArgs &t, const int32_t *aggPtr, const int32_t *dofsByAgg, const double *w, double *rc, hipStream_t s) {
    if (t.dim == 3) hipLaunchKernelGGL(k_tl_restrict<3>, dim3(t.nAgg), dim3(256), 0, s, t, aggPtr, dofsByAgg, w, rc);
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
void launch_tl_rap_agg(const TLArgs &t, const int32_t *aggPtr, const int32_t *dofsByAgg, const int32_t *binCoord, const int32_t *rowPtr,
                       const int32_t *colIdx, const double *vals, double *Ac, hipStream_t s) {
    if (t.dim == 3) hipLaunchKernelGGL(k_tl_rap_agg<3>, dim3(t.nAgg), dim3(512), 0, s, t, aggPtr, dofsByAgg, binCoord, rowPtr, colIdx, vals, Ac);
    else hipLaunchKernelGGL(k_tl_rap_agg<2>, dim3(t.nAgg), dim3(512), 0, s, t, aggPtr, dofsByAgg, binCoord, rowPtr, colIdx, vals, Ac);
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
    if (t.dim == 3) hipLaunchKernelGGL(k_tl_apply<3>, dim3(grid_for(t.nDoF)), dim3(256), 0, s, t, dinv, r, yc, z, scal, it, stopPtr);
    else hipLaunchKernelGGL(k_tl_apply<2>, dim3(grid_for(t.nDoF)), dim3(256), 0, s, t, dinv, r, yc, z, scal, it, stopPtr);
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
    hipLaunchKernelGGL(k_dev_dots, dim3(grid_for(n)), dim3(256), 0, s, n, r, z, out);
    CHECK_LAUNCH();
}

void launch_advance_base(double *stop, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_advance_base, dim3(1), dim3(1), 0, s, stop, (double)n);
    CHECK_LAUNCH();
}

void launch_pcg_direction(int64_t n, const double *z, double *p, const double *scal, int it, const double *stopPtr, hipStream_t s) {
    hipLaunchKernelGGL(k_pcg_direction, dim3(grid_for(n)), dim3(256), 0, s, n, z, p, scal, it, stopPtr);
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
void launch_scatter_values(int64_t n, const int64_t *idx, const double *val, double *v, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_scatter_values, dim3(grid_for(n)), dim3(256), 0, s, n, idx, val, v);
    CHECK_LAUNCH();
}
void launch_dot(int64_t n, const double *a, const double *b, double *out, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_dot, dim3(grid_for(n)), dim3(256), 0, s, n, a, b, out);
    CHECK_LAUNCH();
}

}} // namespace mfh::k
