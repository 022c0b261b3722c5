// Symbolic phase: block-CSR pattern of K and the gather lists that drive the owner-computes
// assembly kernel. This is the analogue of the reference's triplet sort + duplicate merge
// (TripletMatrix::sumRepeated, SparseMatrices.hh:280-374) and CSC build (:422-447), hoisted out
// of the numeric phase: it depends only on connectivity + DoF map and is reused by every
// numeric assembly on the same mesh (the reference redoes the sort on every assembly).
#include "mfh_internal.hh"
#include <numeric>
#include <cmath>

namespace mfh {

void build_symbolic(const HostMesh &m, const std::vector<int32_t> &dofForNode, int64_t nDoF, int64_t nOwnedDoF,
                    int chunkSlots, int contribOrder, bool wantScatter, Symbolic &S, bool upperOnly) {
    const int npe = m.npe;
    const int64_t nElem = m.nElem;
    if ((double)nElem * npe * npe >= 4294967295.0)
        throw Error(MFH_ERR_UNSUPPORTED, "mesh too large for 32-bit contribution codes (partition it across GPUs)");
    S = Symbolic();
    S.nRows = nOwnedDoF;
    S.nCols = nDoF;
    auto dofOf = [&](int64_t e, int i) -> int32_t {
        int32_t node = m.elemNodes[(size_t)e * npe + i];
        return dofForNode.empty() ? node : dofForNode[node];
    };

    // ---- incidence: row -> list of (e*npe + i), element-ascending
    std::vector<int64_t> incPtr((size_t)S.nRows + 1, 0);
    for (int64_t e = 0; e < nElem; ++e)
        for (int i = 0; i < npe; ++i) {
            int32_t d = dofOf(e, i);
            if (d < S.nRows) ++incPtr[(size_t)d + 1];
        }
    for (int64_t r = 0; r < S.nRows; ++r) incPtr[r + 1] += incPtr[r];
    std::vector<uint32_t> inc((size_t)incPtr[S.nRows]);
    {
        std::vector<int64_t> cur(incPtr.begin(), incPtr.end() - 1);
        for (int64_t e = 0; e < nElem; ++e)
            for (int i = 0; i < npe; ++i) {
                int32_t d = dofOf(e, i);
                if (d < S.nRows) inc[(size_t)cur[d]++] = (uint32_t)(e * npe + i);
            }
    }

    // ---- per row: sorted unique columns + contributions grouped by slot (parallel over row ranges
    //      balanced by incidence count)
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), S.nRows / 256 + 1));
    std::vector<int64_t> rangeStart(nt + 1, 0);
    {
        int64_t total = incPtr[S.nRows];
        int64_t r = 0;
        for (int t = 1; t < nt; ++t) {
            int64_t target = total * t / nt;
            while (r < S.nRows && incPtr[r] < target) ++r;
            rangeStart[t] = r;
        }
        rangeStart[nt] = S.nRows;
    }
    struct Part {
        std::vector<int32_t> cols;
        std::vector<uint32_t> code;
        std::vector<uint16_t> slotInRow;
        std::vector<int32_t> rowLen;
        std::vector<int32_t> rowContrib;
    };
    std::vector<Part> parts(nt);
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(nt);
    for (int t = 0; t < nt; ++t) {
        th.emplace_back([&, t] {
            try {
                Part &P = parts[t];
                int64_t r0 = rangeStart[t], r1 = rangeStart[t + 1];
                P.rowLen.reserve(r1 - r0);
                P.rowContrib.reserve(r1 - r0);
                P.code.reserve((size_t)(incPtr[r1] - incPtr[r0]) * npe);
                P.slotInRow.reserve((size_t)(incPtr[r1] - incPtr[r0]) * npe);
                std::vector<uint64_t> pairs;
                for (int64_t r = r0; r < r1; ++r) {
                    pairs.clear();
                    for (int64_t k = incPtr[r]; k < incPtr[r + 1]; ++k) {
                        uint32_t ei = inc[(size_t)k];
                        int64_t e = ei / npe;
                        for (int j = 0; j < npe; ++j) {
                            uint32_t col = (uint32_t)dofOf(e, j);
                            if (upperOnly && (int64_t)col < r) continue;      // upper-only storage: blocks (r, c >= r)
                            pairs.push_back(((uint64_t)col << 32) | (uint64_t)(ei * (uint32_t)npe + (uint32_t)j));
                        }
                    }
                    std::sort(pairs.begin(), pairs.end());
                    int32_t len = 0;
                    uint32_t prev = 0xffffffffu;
                    for (uint64_t pr : pairs) {
                        uint32_t col = (uint32_t)(pr >> 32);
                        if (col != prev) { P.cols.push_back((int32_t)col); prev = col; ++len; }
                        if (len > 65535) throw Error(MFH_ERR_UNSUPPORTED, "row with more than 65535 blocks");
                        P.code.push_back((uint32_t)pr);
                        P.slotInRow.push_back((uint16_t)(len - 1));
                    }
                    P.rowLen.push_back(len);
                    P.rowContrib.push_back((int32_t)pairs.size());
                }
            } catch (...) { err[t] = std::current_exception(); }
        });
    }
    for (auto &x : th) x.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);
    inc.clear(); inc.shrink_to_fit();

    // ---- rowPtr / colIdx
    S.rowPtr.assign((size_t)S.nRows + 1, 0);
    std::vector<int64_t> rowContribPtr((size_t)S.nRows + 1, 0);
    {
        int64_t r = 0, acc = 0, accC = 0;
        for (int t = 0; t < nt; ++t)
            for (size_t k = 0; k < parts[t].rowLen.size(); ++k, ++r) {
                acc += parts[t].rowLen[k];
                accC += parts[t].rowContrib[k];
                if (acc > 2147483647LL) throw Error(MFH_ERR_UNSUPPORTED, "more than 2^31 blocks on one device");
                S.rowPtr[r + 1] = (int32_t)acc;
                rowContribPtr[r + 1] = accC;
                S.maxRowLen = std::max(S.maxRowLen, parts[t].rowLen[k]);
            }
    }
    S.nnzb = S.rowPtr[S.nRows];
    S.colIdx.resize((size_t)S.nnzb);
    {
        size_t off = 0;
        for (int t = 0; t < nt; ++t) {
            std::copy(parts[t].cols.begin(), parts[t].cols.end(), S.colIdx.begin() + off);
            off += parts[t].cols.size();
            std::vector<int32_t>().swap(parts[t].cols);
        }
    }
    S.nMirror = 0;
    for (int64_t r = 0; r < S.nRows; ++r)
        for (int32_t q = S.rowPtr[r]; q < S.rowPtr[r + 1]; ++q) S.nMirror += S.colIdx[q] > r && S.colIdx[q] < S.nRows;

    // ---- chunks
    if (chunkSlots < 64) chunkSlots = 64;
    if (S.maxRowLen > chunkSlots) chunkSlots = ((S.maxRowLen + 63) / 64) * 64;
    if (chunkSlots > 2048)
        throw Error(MFH_ERR_UNSUPPORTED, "a block row has more than 2048 blocks (vertex valence too high for LDS accumulation)");
    S.chunkSlots = chunkSlots;
    S.chunkRow.clear();
    S.chunkRow.push_back(0);
    {
        int64_t r = 0;
        while (r < S.nRows) {
            int32_t s0 = S.rowPtr[r];
            int64_t r2 = r + 1;
            while (r2 < S.nRows && S.rowPtr[r2 + 1] - s0 <= chunkSlots) ++r2;
            S.chunkRow.push_back((int32_t)r2);
            r = r2;
        }
    }
    const int64_t nChunk = S.nChunk();
    // the SpMV kernel streams K and prefers larger chunks (fewer phase switches per byte)
    S.spmvChunkSlots = std::max(512, chunkSlots);
    S.spmvChunkRow.clear();
    S.spmvChunkRow.push_back(0);
    {
        int64_t r = 0;
        while (r < S.nRows) {
            int32_t s0 = S.rowPtr[r];
            int64_t r2 = r + 1;
            while (r2 < S.nRows && S.rowPtr[r2 + 1] - s0 <= S.spmvChunkSlots) ++r2;
            S.spmvChunkRow.push_back((int32_t)r2);
            r = r2;
        }
    }

    // ---- contributions, concatenated in row order, then re-ordered inside every chunk
    const int64_t nContrib = rowContribPtr[S.nRows];
    S.contribCode.resize((size_t)nContrib);
    S.contribSlot.resize((size_t)nContrib);
    std::vector<int64_t> partOff(nt + 1, 0);
    for (int t = 0; t < nt; ++t) partOff[t + 1] = partOff[t] + (int64_t)parts[t].code.size();
    if (wantScatter) S.scatterSlot.assign((size_t)nElem * npe * npe, -1);
    S.contribPtr.resize((size_t)nChunk + 1);
    for (int64_t c = 0; c <= nChunk; ++c) S.contribPtr[c] = rowContribPtr[S.chunkRow[c]];

    // slot relative to chunk start = rowPtr[row] - rowPtr[chunkRow] + slotInRow; rows of a part
    th.clear();
    std::fill(err.begin(), err.end(), nullptr);
    for (int t = 0; t < nt; ++t) {
        th.emplace_back([&, t] {
            try {
                Part &P = parts[t];
                int64_t r0 = rangeStart[t];
                size_t k = 0;
                // chunk of row r0
                int64_t c = std::upper_bound(S.chunkRow.begin(), S.chunkRow.end(), (int32_t)r0) - S.chunkRow.begin() - 1;
                for (size_t ri = 0; ri < P.rowLen.size(); ++ri) {
                    int64_t r = r0 + (int64_t)ri;
                    while (S.chunkRow[c + 1] <= r) ++c;
                    int32_t base = S.rowPtr[r] - S.rowPtr[S.chunkRow[c]];
                    int64_t out = rowContribPtr[r];
                    for (int32_t q = 0; q < P.rowContrib[ri]; ++q, ++k, ++out) {
                        S.contribCode[(size_t)out] = P.code[k];
                        S.contribSlot[(size_t)out] = (uint16_t)(base + P.slotInRow[k]);
                        if (wantScatter) S.scatterSlot[P.code[k]] = S.rowPtr[r] + P.slotInRow[k];
                    }
                }
                Part().code.swap(P.code);
                Part().slotInRow.swap(P.slotInRow);
            } catch (...) { err[t] = std::current_exception(); }
        });
    }
    for (auto &x : th) x.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);

    // ---- ordering inside chunks
    //  0 = rank-major: the k-th contribution of every block before any (k+1)-th: lanes of one wave
    //      hit distinct LDS accumulators (conflict-free ds_add_f64)
    //  1 = element-major: lanes of a wave share the element record (best L1 locality)
    //  2 = slot-major (as produced above)
    if (contribOrder != 2) {
        parallel_ranges(nChunk, [&](int64_t cb, int64_t ce, int) {
            std::vector<uint64_t> keyed;
            std::vector<uint32_t> tmpCode;
            std::vector<uint16_t> tmpSlot;
            for (int64_t c = cb; c < ce; ++c) {
                int64_t b = S.contribPtr[c], e = S.contribPtr[c + 1];
                size_t n = (size_t)(e - b);
                keyed.resize(n);
                if (contribOrder == 0) {
                    uint32_t rank = 0;
                    for (size_t k = 0; k < n; ++k) {
                        if (k > 0 && S.contribSlot[b + k] != S.contribSlot[b + k - 1]) rank = 0;
                        // key: rank (16) | slot (16) | index (32)
                        keyed[k] = ((uint64_t)rank << 48) | ((uint64_t)S.contribSlot[b + k] << 32) | (uint64_t)k;
                        ++rank;
                        if (rank > 65535) throw Error(MFH_ERR_UNSUPPORTED, "block with more than 65535 element contributions");
                    }
                } else {
                    for (size_t k = 0; k < n; ++k) keyed[k] = ((uint64_t)S.contribCode[b + k] << 32) | (uint64_t)k;
                }
                std::sort(keyed.begin(), keyed.end());
                tmpCode.resize(n); tmpSlot.resize(n);
                for (size_t k = 0; k < n; ++k) {
                    size_t src = (size_t)(keyed[k] & 0xffffffffu);
                    tmpCode[k] = S.contribCode[b + src];
                    tmpSlot[k] = S.contribSlot[b + src];
                }
                std::copy(tmpCode.begin(), tmpCode.end(), S.contribCode.begin() + b);
                std::copy(tmpSlot.begin(), tmpSlot.end(), S.contribSlot.begin() + b);
            }
        }, 64);
    }
}

// ------------------------------------------------------------------------------------------------
// Shape-function coefficient tables
// ------------------------------------------------------------------------------------------------
void build_shape_tables(int dim, int deg, ShapeTables &T) {
    const int nv = dim + 1;
    const int npe = nodes_per_elem(dim, deg);
    T.npe = npe;
    // quadrature rule of degree 2(deg-1) on the dim-simplex (GaussQuadrature.hh:115-127, 283-295)
    std::vector<std::array<double, 4>> pts;
    std::vector<double> w;
    if (deg == 1) {
        std::array<double, 4> p{};
        for (int k = 0; k < nv; ++k) p[k] = 1.0 / nv;
        pts.push_back(p); w.push_back(1.0);
    } else if (dim == 3) {
        const double c0 = 0.58541019662496845446, c1 = 0.13819660112501051518;
        for (int q = 0; q < 4; ++q) {
            std::array<double, 4> p{c1, c1, c1, c1};
            p[q] = c0;
            pts.push_back(p); w.push_back(0.25);
        }
    } else {
        const double c0 = 2 / 3.0, c1 = 1 / 6.0;
        for (int q = 0; q < 3; ++q) {
            std::array<double, 4> p{c1, c1, c1, 0};
            p[q] = c0;
            pts.push_back(p); w.push_back(1 / 3.0);
        }
    }
    // grad phi_i(q) = alpha_i(q) gl[s_i] + beta_i(q) gl[t_i]      (EmbeddedElement.hh:315-332)
    auto coef = [&](int i, const std::array<double, 4> &x, double &al, double &be) {
        if (deg == 1) { al = 1.0; be = 0.0; return; }
        if (i < nv) { al = 4.0 * x[i] - 1.0; be = 0.0; return; }
        int e = i - nv;
        al = 4.0 * x[kEdgeEnd[e]];   // multiplies gl[start]
        be = 4.0 * x[kEdgeStart[e]]; // multiplies gl[end]
    };
    for (int i = 0; i < npe; ++i) {
        if (deg == 1 || i < nv) { T.sup_s[i] = i; T.sup_t[i] = i; }
        else { T.sup_s[i] = kEdgeStart[i - nv]; T.sup_t[i] = kEdgeEnd[i - nv]; }
    }
    T.pairTable.assign((size_t)npe * npe * 4, 0.0);
    T.intGrad.assign((size_t)npe * 2, 0.0);
    for (int i = 0; i < npe; ++i)
        for (int j = 0; j < npe; ++j) {
            double acc[4] = {0, 0, 0, 0};
            for (size_t q = 0; q < pts.size(); ++q) {
                double ai, bi, aj, bj;
                coef(i, pts[q], ai, bi);
                coef(j, pts[q], aj, bj);
                acc[0] += w[q] * ai * aj; acc[1] += w[q] * ai * bj;
                acc[2] += w[q] * bi * aj; acc[3] += w[q] * bi * bj;
            }
            for (int k = 0; k < 4; ++k) T.pairTable[((size_t)i * npe + j) * 4 + k] = acc[k];
        }
    // the four distinct pair coefficients the kernels select from (see elem_block), and a check
    // that the selection rule reproduces the whole quadrature table
    if (deg == 2) {
        const int e0 = nv, e1 = nv + 1;   // edge nodes (0,1): A = 4 l_1 grad l_0, B = 4 l_0 grad l_1 ; (1,2): A = 4 l_2 ...
        T.pairConst[0] = T.pairTable[((size_t)0 * npe + 0) * 4];         // vertex 0 x vertex 0         (a == b)
        T.pairConst[1] = T.pairTable[((size_t)0 * npe + 1) * 4];         // vertex 0 x vertex 1         (a != b)
        T.pairConst[2] = T.pairTable[((size_t)1 * npe + e0) * 4];        // vertex 1 x A(edge 01): l=1   (a == b)
        T.pairConst[3] = T.pairTable[((size_t)0 * npe + e0) * 4];        // vertex 0 x A(edge 01): l=1   (a != b)
        T.pairConst[4] = T.pairTable[((size_t)e0 * npe + e0) * 4];       // A(edge 01) x A(edge 01)      (a == b)
        T.pairConst[5] = T.pairTable[((size_t)e0 * npe + e0) * 4 + 1];   // A(edge 01) x B(edge 01)      (a != b)
        (void)e1;
        for (int i = 0; i < npe; ++i)
            for (int j = 0; j < npe; ++j) {
                const bool vi = i < nv, vj = j < nv;
                const int si = T.sup_s[i], ti = T.sup_t[i], sj = T.sup_s[j], tj = T.sup_t[j];
                const int lAi = vi ? si : ti, lBi = si, lAj = vj ? sj : tj, lBj = sj;
                const double *pc = T.pairConst;
                const double c1 = pc[5], c2 = pc[4] - pc[5], c3 = pc[5] - pc[3];
                auto coefS = [&](bool eq, double o, double o2) { return (c1 + (eq ? c2 : 0.0)) - c3 * (o + o2) + o * o2; };
                const double oi = vi ? 1.0 : 0.0, oj = vj ? 1.0 : 0.0;
                double S[4];
                S[0] = coefS(lAi == lAj, oi, oj);
                S[1] = vj ? 0.0 : coefS(lAi == lBj, oi, 0.0);
                S[2] = vi ? 0.0 : coefS(lBi == lAj, 0.0, oj);
                S[3] = (vi || vj) ? 0.0 : coefS(lBi == lBj, 0.0, 0.0);
                for (int k = 0; k < 4; ++k)
                    if (std::fabs(S[k] - T.pairTable[((size_t)i * npe + j) * 4 + k]) > 2e-15)
                        throw Error(MFH_ERR_STATE, "internal: pair-coefficient selection rule does not reproduce the quadrature table");
            }
    }
    // integral of grad phi_i over a unit-volume element: the degree-(deg-1) interpolant is integrated
    // by Interpolant::integrate = vol/(K+1) * sum of vertex values (Functions.hh:246-253)
    for (int i = 0; i < npe; ++i) {
        double al = 0, be = 0;
        if (deg == 1) { al = 1.0; }
        else {
            for (int k = 0; k < nv; ++k) {
                std::array<double, 4> x{0, 0, 0, 0};
                x[k] = 1.0;
                double a, b;
                coef(i, x, a, b);
                al += a / nv; be += b / nv;
            }
        }
        T.intGrad[(size_t)i * 2] = al; T.intGrad[(size_t)i * 2 + 1] = be;
    }
    // reference mass matrix int phi_i phi_j over the unit-volume simplex, exactly: shape functions are
    // polynomials in the barycentric coordinates (Functions.hh:86-102) and
    // int l^a = K! prod(a_k!) / (sum a + K)!  -- what Quadrature<K, 2 Deg> (MassMatrix.hh:66-77) evaluates
    // exactly as well, so the two agree to rounding.
    struct Mono { double c; int e[4]; };
    auto shapePoly = [&](int i) {
        std::vector<Mono> pnl;
        if (deg == 1) { Mono m{1.0, {0, 0, 0, 0}}; m.e[i] = 1; pnl.push_back(m); return pnl; }
        if (i < nv) {   // 2 l_i (l_i - 1/2) = 2 l_i^2 - l_i
            Mono a{2.0, {0, 0, 0, 0}}, b{-1.0, {0, 0, 0, 0}};
            a.e[i] = 2; b.e[i] = 1;
            pnl.push_back(a); pnl.push_back(b);
            return pnl;
        }
        Mono m{4.0, {0, 0, 0, 0}};   // 4 l_s l_t
        m.e[kEdgeStart[i - nv]] += 1; m.e[kEdgeEnd[i - nv]] += 1;
        pnl.push_back(m);
        return pnl;
    };
    auto fact = [](int n) { double f = 1; for (int k = 2; k <= n; ++k) f *= k; return f; };
    T.massRef.assign((size_t)npe * npe, 0.0);
    for (int i = 0; i < npe; ++i)
        for (int j = 0; j < npe; ++j) {
            double acc = 0;
            for (const Mono &a : shapePoly(i))
                for (const Mono &b : shapePoly(j)) {
                    int tot = 0;
                    double num = fact(dim);
                    for (int k = 0; k < nv; ++k) { const int ex = a.e[k] + b.e[k]; tot += ex; num *= fact(ex); }
                    acc += a.c * b.c * num / fact(tot + dim);
                }
            T.massRef[(size_t)i * npe + j] = acc;
        }
}

} // namespace mfh
