// p-multigrid preconditioner for quadratic elasticity (MFH_PRECOND_MULTIGRID): what the PCG that replaces the reference's CHOLMOD
// solve (SPSDSystem::solve, SparseMatrices.hh:2515-2606) uses to reach a fixed, mesh-independent iteration count.
//
//   level 0  quadratic mesh            operator: matrix-free cluster kernel (k_mf_cluster)      smoother: Chebyshev on D0^-1 K
//   level 1  linear mesh, same vertices operator: assembled block-CSR K1 (k_spmv)                smoother: Chebyshev on D1^-1 K1
//   level 2  rigid-body modes of ~1000 geometric aggregates: dense inverse (the coarse level of the two-level preconditioner)
//
// P1 is a subspace of P2 (phi^P1_v = phi^P2_v + 1/2 sum_{edges e at v} phi^P2_e), so the Galerkin operator P^T K2 P IS the linear
// stiffness matrix of the same elements: level 1 is a second context of this library (`coarse`) that assembles K1 with the same
// kernels from the same element records; nothing is multiplied out. One application z = M^-1 r is a symmetric V-cycle (the same
// Chebyshev polynomial before and after the coarse correction on both levels), so M is SPD and plain PCG applies.
#include "mfh_ctx.hh"

namespace mfhi {

k::TLArgs tl_args(mfh_ctx *c);
void apply_operator(mfh_ctx *c, bool masked, const double *x, double *y, double *dotOut);
void ensure_fixed_uploaded(mfh_ctx *c);
double device_dot(mfh_ctx *c, int64_t n, const double *a, const double *b);
void upload_mesh(mfh_ctx *c);

namespace {

// largest eigenvalue of D^-1 K on the free variables: power iteration from a pseudo-random start, all on the device. The iterate is
// not normalised (lambda_max <= dim + 1 for a block-Jacobi-scaled stiffness matrix: 16 steps grow it by < 1e10) and only the last
// two norms are read back: one host synchronisation.
double estimate_lambda_max(mfh_ctx *c, DBuf<double> &v, DBuf<double> &w, DBuf<double> &t) {
    const int d = c->bs();
    const int64_t nRows = c->sym.nRows, n = nRows * d;
    hipStream_t s = c->stream;
    k::launch_fill_hash(n, v.p, s);
    const bool masked = !c->fixedVars.empty();
    if (masked) k::launch_mask(n, c->dFixedMask.p, v.p, s);
    double *a = v.p, *b = w.p;
    const int steps = 16;
    for (int it = 0; it < steps; ++it) {
        apply_operator(c, masked, a, t.p, nullptr);                         // t = K a
        // b = D^-1 t  (k_mg_cheb with first + assign: d = D^-1 rin, x = d)
        k::launch_mg_cheb(d, nRows, c->dDinv.p, t.p, nullptr, nullptr, b, b, 0.0, 1.0, true, true, nullptr, 0, nullptr, s);
        std::swap(a, b);
    }
    // a = (D^-1 K)^steps v, b = the iterate before it
    c->stop.alloc(4);
    MFH_HIP(hipMemsetAsync(c->stop.p + 1, 0, 2 * sizeof(double), s));
    k::launch_dot(n, a, a, c->stop.p + 1, s);
    k::launch_dot(n, b, b, c->stop.p + 2, s);
    double h[2] = {0, 0};
    MFH_HIP(hipMemcpyAsync(h, c->stop.p + 1, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    return h[1] > 0 ? std::sqrt(h[0] / h[1]) : 1.0;
}

struct Level {   // what a Chebyshev sweep needs of a level
    mfh_ctx *c;
    double lmax, ratio;
    int steps;
};

// x = S(b) (zeroInit) or x <- S(b, x): `steps` Chebyshev steps on D^-1 K over [lmax ratio, lmax]. r, dvec, t: work vectors.
// On return (wantResidual) `r` and `t` are such that the residual b - K x equals r - t (the restriction subtracts on the fly).
void chebyshev(const Level &L, const double *b, double *x, bool zeroInit, bool wantResidual, double *r, double *dvec, double *t,
               const double *scal, int it, const double *stop) {
    mfh_ctx *c = L.c;
    const int d = c->bs();
    const int64_t nRows = c->sym.nRows;
    hipStream_t s = c->stream;
    const bool masked = !c->fixedVars.empty();
    const double lmin = L.lmax * L.ratio, theta = 0.5 * (L.lmax + lmin), delta = 0.5 * (L.lmax - lmin), sigma = theta / delta;
    double rho = 1.0 / sigma;
    const double *rin = b;
    const double *tin = nullptr;
    if (!zeroInit) {            // r = b - K x
        apply_operator(c, masked, x, t, nullptr);
        tin = t;
    }
    // step 1: d = D^-1 r / theta ; x (+)= d ; the running residual goes to r
    k::launch_mg_cheb(d, nRows, c->dDinv.p, rin, tin, r, dvec, x, 0.0, 1.0 / theta, true, zeroInit, scal, it, stop, s);
    for (int j = 1; j < L.steps; ++j) {
        const double rhoNew = 1.0 / (2.0 * sigma - rho);
        apply_operator(c, masked, dvec, t, nullptr);                        // r -= K d inside the next step
        k::launch_mg_cheb(d, nRows, c->dDinv.p, r, t, r, dvec, x, rhoNew * rho, 2.0 * rhoNew / delta, false, false, scal, it, stop, s);
        rho = rhoNew;
    }
    if (wantResidual) apply_operator(c, masked, dvec, t, nullptr);          // residual = r - K d_last
}

}   // namespace

void destroy_multigrid(mfh_ctx *c) {
    auto &G = c->mg;
    G.valid = false;
    if (G.coarse) { mfh_destroy(G.coarse); G.coarse = nullptr; }
}

// Builds (or rebuilds) the hierarchy. false (with a note in precondNote) when it does not apply; the caller then falls back.
bool ensure_multigrid(mfh_ctx *c) {
    auto &G = c->mg;
    if (G.valid) return true;
    c->precondNote.clear();
    const HostMesh &m = c->mesh;
    if (c->op != MFH_OP_ELASTICITY || c->external || !c->haveMesh || m.deg != 2) {
        c->precondNote = "p-multigrid needs quadratic elasticity elements: using the two-level preconditioner";
        return false;
    }
    if (c->sym.nRows != c->sym.nCols) {
        c->precondNote = "p-multigrid unavailable for partitioned rows: using block-Jacobi";
        return false;
    }
    const double t0 = now_ms();
    destroy_multigrid(c);
    hipStream_t s = c->stream;
    const int d = m.dim, nv = d + 1, npe = m.npe;
    const int64_t nDoF = c->nDoF;
    const bool timing = getenv("MFH_MG_TIMING") != nullptr;
    double tp = now_ms();
    auto lap = [&](const char *what) {
        if (!timing) return;
        MFH_HIP(hipStreamSynchronize(s));
        const double t = now_ms();
        fprintf(stderr, "[multigrid setup] %-34s %8.2f ms\n", what, t - tp);
        tp = t;
    };
    // ---- vertex nodes (the first dim + 1 nodes of every element) -> coarse nodes, coarse DoFs; parents of every fine DoF.
    // Element loops run on the host threads; several elements may store the SAME value to one entry (relaxed atomic stores).
    std::vector<int32_t> coarseNode((size_t)m.nNode, -1);
    parallel_ranges(m.nElem, [&](int64_t eb, int64_t ee, int) {
        for (int64_t e = eb; e < ee; ++e)
            for (int k2 = 0; k2 < nv; ++k2) __atomic_store_n(&coarseNode[(size_t)m.elemNodes[(size_t)e * npe + k2]], 0, __ATOMIC_RELAXED);
    });
    int64_t nCN = 0;
    for (int64_t n = 0; n < m.nNode; ++n)
        if (coarseNode[(size_t)n] == 0) coarseNode[(size_t)n] = (int32_t)nCN++;
    std::vector<int32_t> coarseDofOfFine((size_t)nDoF, -1);
    int64_t nCD = 0;
    for (int64_t n = 0; n < m.nNode; ++n) {       // coarse DoFs numbered in node order, like applyPeriodicConditions numbers DoFs
        if (coarseNode[(size_t)n] < 0) continue;
        const int32_t f = dof_of(c, n);
        if (coarseDofOfFine[(size_t)f] < 0) coarseDofOfFine[(size_t)f] = (int32_t)nCD++;
    }
    std::vector<int32_t> parA((size_t)nDoF, -1), parB((size_t)nDoF, -1), fineOf((size_t)nCD, -1);
    parallel_ranges(nDoF, [&](int64_t fb, int64_t fe, int) {
        for (int64_t f = fb; f < fe; ++f)
            if (coarseDofOfFine[(size_t)f] >= 0) { parA[(size_t)f] = parB[(size_t)f] = coarseDofOfFine[(size_t)f]; fineOf[(size_t)coarseDofOfFine[(size_t)f]] = (int32_t)f; }
    });
    parallel_ranges(m.nElem, [&](int64_t eb, int64_t ee, int) {
        for (int64_t e = eb; e < ee; ++e) {
            const int32_t *en = &m.elemNodes[(size_t)e * npe];
            for (int k2 = nv; k2 < npe; ++k2) {
                const int32_t f = dof_of(c, en[k2]);
                if (coarseDofOfFine[(size_t)f] >= 0) continue;      // (a periodic image that is a vertex elsewhere keeps the vertex rule)
                const int a = d == 3 ? kEdgeStart[k2 - nv] : (k2 - nv), b = d == 3 ? kEdgeEnd[k2 - nv] : ((k2 - nv + 1) % 3);
                int32_t pa = coarseDofOfFine[(size_t)dof_of(c, en[a])], pb = coarseDofOfFine[(size_t)dof_of(c, en[b])];
                if (pa > pb) std::swap(pa, pb);                     // every element names the two ends in the same order
                __atomic_store_n(&parA[(size_t)f], pa, __ATOMIC_RELAXED);
                __atomic_store_n(&parB[(size_t)f], pb, __ATOMIC_RELAXED);
            }
        }
    });
    lap("vertices, coarse DoFs, parents");
    // restriction lists: coarse DoF -> the edge-node DoFs it is an end of
    std::vector<int32_t> resPtr((size_t)nCD + 1, 0);
    for (int64_t f = 0; f < nDoF; ++f)
        if (parA[(size_t)f] >= 0 && coarseDofOfFine[(size_t)f] < 0) { ++resPtr[(size_t)parA[(size_t)f] + 1]; ++resPtr[(size_t)parB[(size_t)f] + 1]; }
    for (int64_t q = 0; q < nCD; ++q) resPtr[(size_t)q + 1] += resPtr[(size_t)q];
    std::vector<int32_t> resIdx((size_t)resPtr[(size_t)nCD]), cur(resPtr.begin(), resPtr.end() - 1);
    for (int64_t f = 0; f < nDoF; ++f)
        if (parA[(size_t)f] >= 0 && coarseDofOfFine[(size_t)f] < 0) {
            resIdx[(size_t)cur[(size_t)parA[(size_t)f]]++] = (int32_t)f;
            resIdx[(size_t)cur[(size_t)parB[(size_t)f]]++] = (int32_t)f;
        }
    lap("restriction lists");
    // ---- level 1: a context of its own on the vertices (degree 1), sharing device and stream
    mfh_ctx *c1 = new mfh_ctx();
    G.coarse = c1;
    c1->device = c->device; c1->stream = c->stream; c1->ownStream = false; c1->nCU = c->nCU;
    c1->symbolicDevice = c->symbolicDevice; c1->topologyDevice = c->topologyDevice;
    HostMesh &m1 = c1->mesh;
    m1 = HostMesh();
    m1.dim = d; m1.deg = 1; m1.npe = nv; m1.npbe = nodes_per_bdry_elem(d, 1);
    m1.nElem = m.nElem; m1.nNode = nCN; m1.nVert = nCN; m1.nOwned = nCN;
    m1.elemNodes.resize((size_t)m.nElem * nv);
    parallel_ranges(m.nElem, [&](int64_t eb, int64_t ee, int) {
        for (int64_t e = eb; e < ee; ++e)
            for (int k2 = 0; k2 < nv; ++k2) m1.elemNodes[(size_t)e * nv + k2] = coarseNode[(size_t)m.elemNodes[(size_t)e * npe + k2]];
    });
    m1.nodePos.resize((size_t)nCN * d);
    parallel_ranges(m.nNode, [&](int64_t nb, int64_t ne, int) {
        for (int64_t n = nb; n < ne; ++n)
            if (coarseNode[(size_t)n] >= 0)
                for (int a = 0; a < d; ++a) m1.nodePos[(size_t)coarseNode[(size_t)n] * d + a] = m.nodePos[(size_t)n * d + a];
    });
    m1.vertPos = m1.nodePos;
    m1.isBdryNode.assign((size_t)nCN, 0);
    upload_mesh(c1);
    lap("linear mesh + upload");
    // material: the same per-element parameters (k_geometry rebuilds the records of the linear elements from them)
    c1->matMode = c->matMode; c1->matKind = c->matKind; c1->matParams = c->matParams;
    c1->geoValid = false;
    // DoF map of the vertices (periodic identifications carry over)
    if (nCD != nCN) {
        c1->dofForNode.assign((size_t)nCN, 0);
        for (int64_t n = 0; n < m.nNode; ++n)
            if (coarseNode[(size_t)n] >= 0) c1->dofForNode[(size_t)coarseNode[(size_t)n]] = coarseDofOfFine[(size_t)dof_of(c, n)];
        c1->nDoF = nCD;
        c1->dofUploaded = false;
    }
    // fixed variables of the vertices (homogeneous: the preconditioner acts on corrections)
    {
        std::vector<int64_t> fv;
        for (int64_t v : c->fixedVars) {
            const int64_t f = v / d;
            if (coarseDofOfFine[(size_t)f] >= 0) fv.push_back((int64_t)coarseDofOfFine[(size_t)f] * d + v % d);
        }
        clear_fixed(c1);
        if (!fv.empty()) add_fixed(c1, (int64_t)fv.size(), fv.data(), nullptr);
    }
    c1->matrixFree = 0;                         // the linear level multiplies by its assembled matrix
    c1->matrixStorage = 0;
    c1->aggNodes = c->mgAggNodes;
    c1->precond = MFH_PRECOND_TWO_LEVEL;
    ensure_precond(c1);
    lap("linear level: symbolic + assembly");
    const bool haveCoarse = ensure_twolevel(c1);
    lap("linear level: rigid-body coarse");
    // ---- device copies, work vectors
    G.nFine = nDoF; G.nCoarse = nCD;
    G.parA.upload(parA, s); G.parB.upload(parB, s); G.fineOf.upload(fineOf, s); G.resPtr.upload(resPtr, s);
    G.resIdx.upload(resIdx.empty() ? std::vector<int32_t>{0} : resIdx, s);
    const int64_t n0 = (int64_t)d * nDoF, n1 = (int64_t)d * nCD;
    G.r0.alloc((size_t)n0); G.d0.alloc((size_t)n0); G.t0.alloc((size_t)n0);
    G.b1.alloc((size_t)n1); G.x1.alloc((size_t)n1); G.r1.alloc((size_t)n1); G.d1.alloc((size_t)n1); G.t1.alloc((size_t)n1);
    // ---- spectra of the two Jacobi-preconditioned operators
    ensure_precond(c);
    // the power iteration approaches the largest eigenvalue from below: a margin keeps the Chebyshev polynomials bounded on the whole spectrum
    G.lmax0 = c->mgEigMargin * estimate_lambda_max(c, G.r0, G.d0, G.t0);
    G.lmax1 = c->mgEigMargin * estimate_lambda_max(c1, G.r1, G.d1, G.t1);
    lap("uploads + eigenvalue estimates");
    G.rigidCoarse = haveCoarse;
    G.setup_ms = now_ms() - t0;
    G.valid = true;
    if (!haveCoarse) c->precondNote = "p-multigrid: no rigid-body coarse level (" + c1->precondNote + "); the linear level is smoothed only";
    return true;
}

// z = M^-1 r: one symmetric V-cycle. scal / it / stop: the gate of the PCG iteration this application belongs to (null: none).
void mg_precond(mfh_ctx *c, const double *r, double *z, const double *scal, int it, const double *stop) {
    auto &G = c->mg;
    mfh_ctx *c1 = G.coarse;
    hipStream_t s = c->stream;
    const int d = c->bs();
    const Level L0{c, G.lmax0, c->mgRatio0, c->mgSteps0}, L1{c1, G.lmax1, c->mgRatio1, c->mgSteps1};
    const bool masked0 = !c->fixedVars.empty(), masked1 = !c1->fixedVars.empty();
    // level 0, pre-smoothing from zero; residual r - K z = r0 - t0
    chebyshev(L0, r, z, true, true, G.r0.p, G.d0.p, G.t0.p, scal, it, stop);
    k::launch_mg_restrict(d, G.nCoarse, G.fineOf.p, G.resPtr.p, G.resIdx.p, G.r0.p, G.t0.p, masked1 ? c1->dFixedMask.p : nullptr, G.b1.p, scal, it, stop, s);
    // level 1: (smooth, rigid-body correction, smooth) x mgCoarseCycles -- repeating one symmetric stationary iteration keeps M symmetric
    for (int cyc = 0; cyc < c->mgCoarseCycles; ++cyc) {
        chebyshev(L1, G.b1.p, G.x1.p, cyc == 0, G.rigidCoarse, G.r1.p, G.d1.p, G.t1.p, scal, it, stop);
        if (G.rigidCoarse) {
            auto &T = c1->tl;
            const k::TLArgs ta = tl_args(c1);
            k::launch_mg_diff((int64_t)d * G.nCoarse, G.r1.p, G.t1.p, G.r1.p, scal, it, stop, s);        // residual of level 1
            k::launch_tl_restrict(ta, T.aggPtr.p, T.dofsByAgg.p, G.r1.p, T.rc.p, s);
            k::launch_tl_gemv(T.m, T.ldInv, T.Ainv.p, T.rc.p, T.yc.p, s);
            k::launch_mg_tl_prolong_add(ta, T.yc.p, G.x1.p, scal, it, stop, s);
        }
        chebyshev(L1, G.b1.p, G.x1.p, false, false, G.r1.p, G.d1.p, G.t1.p, scal, it, stop);
    }
    // back to level 0: z += P x1, post-smoothing
    k::launch_mg_prolong_add(d, G.nFine, G.parA.p, G.parB.p, G.x1.p, masked0 ? c->dFixedMask.p : nullptr, z, scal, it, stop, s);
    chebyshev(L0, r, z, false, false, G.r0.p, G.d0.p, G.t0.p, scal, it, stop);
}

}   // namespace mfhi
