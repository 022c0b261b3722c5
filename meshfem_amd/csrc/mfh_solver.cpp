// The PCG of libmeshfem_hip: Chronopoulos-Gear form (one reduction point per iteration) on NR interleaved right-hand sides,
// the same loop for a whole mesh on one GPU and for a row-partitioned mesh on N GPUs (SPSDSystem::solve,
// SparseMatrices.hh:2515-2606, with CHOLMOD replaced; the reference has no distributed code).
//
//   u = M^-1 r, w = K u, gamma = (r, u), delta = (w, u)                         -- ONE all-reduce of {gamma, delta, rr} x NR
//   beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old)
//   p = u + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s               -- one fused vector kernel (k_cg_update)
//
// Multi-GPU: rows (nodes) are partitioned, every rank holds the elements of its owned nodes (mfh_mesh_set with nOwned);
// per iteration the owned entries of u that other ranks read are packed by a kernel into persistent send buffers and
// exchanged on a second HIP stream WHILE the element blocks / row chunks that touch no halo column are processed; the
// others follow once the halo has arrived. Communication goes through an mfh_comm: RCCL (looked up with dlopen, so that
// the library links against no particular ROCm communication stack) or two caller-supplied callbacks.
#include "mfh_ctx.hh"
#include "mfh_comm.hh"
#include <dlfcn.h>
#include <chrono>
#include <thread>

// ------------------------------------------------------------------------------------------------ communicator
namespace {

typedef int (*nccl_get_unique_id_t)(void *);
typedef int (*nccl_comm_init_rank_t)(void **, int, mfh_rccl_unique_id, int);
typedef int (*nccl_comm_destroy_t)(void *);
typedef int (*nccl_all_reduce_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_send_t)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_recv_t)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*nccl_group_t)(void);
typedef const char *(*nccl_error_string_t)(int);
typedef int (*nccl_get_version_t)(int *);

struct RcclApi {
    bool tried = false, ok = false;
    std::string where;
    nccl_get_unique_id_t getUniqueId = nullptr;
    nccl_comm_init_rank_t commInitRank = nullptr;
    nccl_comm_destroy_t commDestroy = nullptr, commAbort = nullptr;
    nccl_all_reduce_t allReduce = nullptr;
    nccl_send_t send = nullptr;
    nccl_recv_t recv = nullptr;
    nccl_group_t groupStart = nullptr, groupEnd = nullptr;
    nccl_error_string_t errorString = nullptr;
    nccl_get_version_t getVersion = nullptr;
};

// RCCL through dlopen: a process that already has an RCCL (PyTorch ships its own librccl.so) keeps using that one; a plain
// C++ client gets the ROCm installation's. No link-time dependency.
RcclApi &rccl() {
    static RcclApi api;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (api.tried) return api;
    api.tried = true;
    void *h = nullptr;
    const char *env = getenv("MFH_RCCL_LIB");
    if (env && *env) { h = dlopen(env, RTLD_NOW | RTLD_LOCAL); api.where = env; }
    if (!h && dlsym(RTLD_DEFAULT, "ncclAllReduce")) { h = RTLD_DEFAULT; api.where = "already loaded in the process"; }
    if (!h) {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL);
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) { api.where = name; break; }
        }
    }
    if (!h) return api;
    auto sym = [&](const char *n) { return dlsym(h, n); };
    api.getUniqueId = (nccl_get_unique_id_t)sym("ncclGetUniqueId");
    api.commInitRank = (nccl_comm_init_rank_t)sym("ncclCommInitRank");
    api.commDestroy = (nccl_comm_destroy_t)sym("ncclCommDestroy");
    api.commAbort = (nccl_comm_destroy_t)sym("ncclCommAbort");
    api.allReduce = (nccl_all_reduce_t)sym("ncclAllReduce");
    api.send = (nccl_send_t)sym("ncclSend");
    api.recv = (nccl_recv_t)sym("ncclRecv");
    api.groupStart = (nccl_group_t)sym("ncclGroupStart");
    api.groupEnd = (nccl_group_t)sym("ncclGroupEnd");
    api.errorString = (nccl_error_string_t)sym("ncclGetErrorString");
    api.getVersion = (nccl_get_version_t)sym("ncclGetVersion");
    api.ok = api.getUniqueId && api.commInitRank && api.commDestroy && api.allReduce && api.send && api.recv && api.groupStart && api.groupEnd;
    return api;
}

constexpr int kNcclFloat64 = 8, kNcclSum = 0;   // rccl.h: ncclDataType_t / ncclRedOp_t

}   // namespace

namespace {

void rccl_check(int rc, const char *what) {
    if (rc == 0) return;
    RcclApi &a = rccl();
    throw Error(MFH_ERR_HIP, std::string(what) + ": " + (a.errorString ? a.errorString(rc) : "RCCL error " + std::to_string(rc)));
}

}   // namespace

namespace mfh {
// the transport underneath: RCCL or the caller's callbacks
void base_allreduce(mfh_comm *cm, double *dev, int64_t n, hipStream_t s) {
    if (!cm || cm->world <= 1 || n == 0) return;
    if (cm->aborted) throw Error(MFH_ERR_HIP, "the RCCL communicator was aborted by a timed-out self test");
    if (cm->nccl) rccl_check(rccl().allReduce(dev, dev, (size_t)n, kNcclFloat64, kNcclSum, cm->nccl, s), "ncclAllReduce");
    else if (cm->allreduce(cm->user, dev, n, (void *)s) != MFH_OK) throw Error(MFH_ERR_HIP, "communicator callback allreduce_sum failed");
}

void base_exchange(mfh_comm *cm, int nPeers, const int32_t *peers, const double *const *sendBufs, const int64_t *sendCounts,
                   double *const *recvBufs, const int64_t *recvCounts, hipStream_t s) {
    if (!cm || cm->world <= 1 || nPeers == 0) return;
    if (cm->aborted) throw Error(MFH_ERR_HIP, "the RCCL communicator was aborted by a timed-out self test");
    if (cm->nccl) {
        RcclApi &a = rccl();
        rccl_check(a.groupStart(), "ncclGroupStart");
        for (int k = 0; k < nPeers; ++k) {
            if (sendCounts[k] > 0) rccl_check(a.send(sendBufs[k], (size_t)sendCounts[k], kNcclFloat64, peers[k], cm->nccl, s), "ncclSend");
            if (recvCounts[k] > 0) rccl_check(a.recv(recvBufs[k], (size_t)recvCounts[k], kNcclFloat64, peers[k], cm->nccl, s), "ncclRecv");
        }
        rccl_check(a.groupEnd(), "ncclGroupEnd");
    } else if (cm->exchange(cm->user, nPeers, peers, sendBufs, sendCounts, recvBufs, recvCounts, (void *)s) != MFH_OK)
        throw Error(MFH_ERR_HIP, "communicator callback exchange failed");
}
}   // namespace mfh

namespace {

// what the solver calls: direct device-to-device transfers (mfh_peer.hip) when the communicator has them and the message fits, the
// transport underneath otherwise. Both decisions come out the same on every rank (peer_can_*).
void comm_allreduce(mfh_comm *cm, double *dev, int64_t n, hipStream_t s) {
    if (!cm || cm->world <= 1 || n == 0) return;
    if (peer_can_allreduce(cm, n)) peer_allreduce(cm, dev, n, s);
    else { base_allreduce(cm, dev, n, s); if (cm->peer.enabled) ++cm->peer.fallbackAllreduces; }
}

void comm_exchange(mfh_comm *cm, int nPeers, const int32_t *peers, const double *const *sendBufs, const int64_t *sendCounts,
                   double *const *recvBufs, const int64_t *recvCounts, hipStream_t s, int *transport = nullptr, bool widthReserved = true) {
    if (!cm || cm->world <= 1 || nPeers == 0) return;
    // widthReserved: the doubles per block row do not exceed what mfh_dist_setup registered with the staging (bs: one right-hand side) -- the same on every rank,
    // unlike the byte counts of a particular pair
    if (widthReserved && peer_can_exchange(cm, nPeers, peers, sendCounts, recvCounts)) {
        peer_exchange(cm, nPeers, peers, sendBufs, sendCounts, recvBufs, recvCounts, s);
        if (transport) *transport = 2;
    } else {
        base_exchange(cm, nPeers, peers, sendBufs, sendCounts, recvBufs, recvCounts, s);
        if (cm->peer.enabled) ++cm->peer.fallbackExchanges;
        if (transport) *transport = cm->nccl ? 1 : 3;
    }
}

}   // namespace

// ------------------------------------------------------------------------------------------------ solver
namespace mfhi {

// the context stops referring to its communicator (mesh replaced, context destroyed)
void dist_detach(mfh_ctx *c) {
    mfh_comm *cm = c->dist.comm;
    if (cm) {
        std::lock_guard<std::mutex> lock(cm->mu);
        cm->users.erase(std::remove(cm->users.begin(), cm->users.end(), c), cm->users.end());
    }
    c->dist.comm = nullptr;
    c->dist.listKind = 0;
}

k::SpmvArgs spmv_args(mfh_ctx *c, bool masked);
k::SpmvMfArgs spmv_mf_cluster_args(mfh_ctx *c, bool masked);
bool prepare_matrix_free(mfh_ctx *c);
k::TLArgs tl_args(mfh_ctx *c);
void ensure_fixed_uploaded(mfh_ctx *c);
void solve_one_classic(mfh_ctx *c, const double *f, double *u, double rtol, int maxit, mfh_solve_info *info);
void apply_operator(mfh_ctx *c, bool masked, const double *x, double *y, double *dotOut);

namespace {

struct DistLink {   // stream / event plumbing of one solve on a partitioned context
    mfh_ctx *c;
    mfh_comm *cm;
    hipStream_t s, cs;
    bool active;
    explicit DistLink(mfh_ctx *c_) : c(c_), cm(c_->dist.comm), s(c_->stream), cs(c_->dist.commStream), active(c_->dist.comm && c_->dist.comm->world > 1) {}
    // owned entries -> persistent send buffers -> neighbours; returns with the exchange in flight on the communication stream
    void halo_begin(double *v, int W) {
        if (!active) return;
        auto &D = c->dist;
        const int64_t nSend = D.sendPtr.back();
        D.sendBuf.alloc((size_t)std::max<int64_t>(1, nSend) * D.sendBufW);
        if (W > D.sendBufW) throw Error(MFH_ERR_STATE, "halo send buffer too small");
        k::launch_pack_rows(nSend, W, D.sendIdx.p, v, D.sendBuf.p, s);
        MFH_HIP(hipEventRecord(D.ev[0], s));
        MFH_HIP(hipStreamWaitEvent(cs, D.ev[0], 0));
        const int np = (int)D.peers.size();
        std::vector<const double *> sb((size_t)np);
        std::vector<double *> rb((size_t)np);
        std::vector<int64_t> sc((size_t)np), rc((size_t)np);
        const int64_t nRows = c->sym.nRows;
        for (int k = 0; k < np; ++k) {
            sb[k] = D.sendBuf.p + D.sendPtr[k] * W;
            sc[k] = (D.sendPtr[k + 1] - D.sendPtr[k]) * W;
            rb[k] = v + (nRows + D.recvPtr[k]) * W;
            rc[k] = (D.recvPtr[k + 1] - D.recvPtr[k]) * W;
        }
        Dist::Profile *pf = profile_slot();
        if (pf) MFH_HIP(hipEventRecord(pf->ev[1], cs));
        comm_exchange(cm, np, D.peers.data(), sb.data(), sc.data(), rb.data(), rc.data(), cs, &D.transport, W <= c->bs());
        MFH_HIP(hipEventRecord(D.ev[1], cs));
        if (pf) MFH_HIP(hipEventRecord(pf->ev[2], cs));
        ++D.nExchanges;
    }
    void halo_end() {
        if (active) MFH_HIP(hipStreamWaitEvent(s, c->dist.ev[1], 0));
    }
    // option "dist_profile": the first applications of the operator in a solve are bracketed by timed events (read at the end of the solve)
    using Dist = mfh_ctx::Dist;
    Dist::Profile *profile_slot() {
        auto &D = c->dist;
        if (!D.profile || D.profOpen < 0) return nullptr;
        return &D.prof[(size_t)D.profOpen];
    }
    void profile_begin() {
        auto &D = c->dist;
        D.profOpen = -1;
        if (!active || !D.profile || D.profUsed >= (int)D.prof.size()) return;
        D.profOpen = D.profUsed++;
        auto &P = D.prof[(size_t)D.profOpen];
        if (!P.ev[0]) for (auto &e : P.ev) MFH_HIP(hipEventCreate(&e));
        MFH_HIP(hipEventRecord(P.ev[0], s));
    }
    void profile_mark(int k) {
        if (Dist::Profile *pf = profile_slot()) MFH_HIP(hipEventRecord(pf->ev[k], s));
    }
    void profile_end() { profile_mark(4); c->dist.profOpen = -1; }
    // in-place sum over the ranks, ordered after everything enqueued on the compute stream so far
    void allreduce(double *dev, int64_t n) {
        if (!active) return;
        auto &D = c->dist;
        MFH_HIP(hipEventRecord(D.ev[2], s));
        MFH_HIP(hipStreamWaitEvent(cs, D.ev[2], 0));
        comm_allreduce(cm, dev, n, cs);
        MFH_HIP(hipEventRecord(D.ev[3], cs));
        MFH_HIP(hipStreamWaitEvent(s, D.ev[3], 0));
    }
};

bool cluster_operator(mfh_ctx *c) { return c->use_mf() && c->mfModeEff() == 4 && c->op == MFH_OP_ELASTICITY; }

// element blocks (cluster operator) / row chunks (assembled SpMV) ordered interior first, then those reading a halo column
void ensure_overlap_lists(mfh_ctx *c, bool cluster) {
    auto &D = c->dist;
    const int kind = cluster ? 1 : 2;
    if (D.listKind == kind && D.listGen == c->listsGen) return;
    hipStream_t s = c->stream;
    const int64_t n = cluster ? c->mfc.nBlocks : (int64_t)c->sym.spmvChunkRow.size() - 1;
    DBuf<uint8_t> flag;
    flag.alloc((size_t)std::max<int64_t>(1, n));
    if (cluster) k::launch_flag_halo_blocks(n, c->mfcDev.blockPtr.p, c->mfcDev.entryDest.p, flag.p, s);
    else k::launch_flag_halo_chunks(n, c->dSpmvChunkRow.p, c->dRowPtr.p, c->dColIdx.p, c->sym.nRows, flag.p, s);
    std::vector<uint8_t> h((size_t)n);
    flag.download(h.data(), h.size(), s);
    std::vector<int32_t> list;
    list.reserve((size_t)n);
    for (int64_t b = 0; b < n; ++b) if (!h[(size_t)b]) list.push_back((int32_t)b);
    D.nInterior = (int64_t)list.size();
    for (int64_t b = 0; b < n; ++b) if (h[(size_t)b]) list.push_back((int32_t)b);
    D.nBoundary = n - D.nInterior;
    D.opList.upload(list.empty() ? std::vector<int32_t>{0} : list, s);
    D.listKind = kind;
    D.listGen = c->listsGen;
}

// y = K x for NR interleaved vectors. With an active communicator the halo part of x is exchanged first, overlapped with
// the interior blocks / chunks. dotOut (stride 4) or the PCG bookkeeping (scal, it, ctl) as in the kernels.
void apply_op_nr(mfh_ctx *c, DistLink &L, int NR, double *x, double *y, bool masked, double *dotOut, double *scal, int it, const double *ctl,
                 int pcgMode = 0, bool smoother = false) {
    hipStream_t s = c->stream;
    const bool cluster = cluster_operator(c);
    const int W = NR * c->bs();
    if (cluster) {
        k::SpmvMfArgs a = spmv_mf_cluster_args(c, masked);
        a.pcgMode = pcgMode;
        if ((size_t)std::max<int64_t>(c->mfc.nIface, 1) * W > c->mfcDev.ifaceBuf.n) throw Error(MFH_ERR_STATE, "interface buffer too small for this batch");
        if (L.active) {
            ensure_overlap_lists(c, true);
            L.profile_begin();
            L.halo_begin(x, W);
            k::launch_mf_cluster_nr(a, NR, x, y, dotOut, scal, it, ctl, c->dist.opList.p, c->dist.nInterior, s);
            L.profile_mark(3);
            L.halo_end();
            k::launch_mf_cluster_nr(a, NR, x, y, dotOut, scal, it, ctl, c->dist.opList.p + c->dist.nInterior, c->dist.nBoundary, s);
            L.profile_end();
        } else
            k::launch_mf_cluster_nr(a, NR, x, y, dotOut, scal, it, ctl, nullptr, c->mfc.nBlocks, s);
        k::launch_mf_rows_nr(a, NR, x, y, dotOut, scal, it, ctl, s);
    } else {
        require_full_storage(c, "the assembled SpMV of the PCG");
        k::SpmvArgs a = spmv_args(c, masked);
        a.pcgMode = pcgMode;
        // inside the multigrid preconditioner (dist_apply) the assembled matrix is read from its FP32 copy when the hierarchy made one (mg_coarse_fp32)
        if (smoother && c->dVals32.p && c->dVals32.n == c->dVals.n) a.vals32 = c->dVals32.p;
        if (L.active) {
            ensure_overlap_lists(c, false);
            L.profile_begin();
            L.halo_begin(x, W);
            k::launch_spmv_nr(a, NR, x, y, dotOut, scal, it, ctl, c->dist.opList.p, c->dist.nInterior, s);
            L.profile_mark(3);
            L.halo_end();
            k::launch_spmv_nr(a, NR, x, y, dotOut, scal, it, ctl, c->dist.opList.p + c->dist.nInterior, c->dist.nBoundary, s);
            L.profile_end();
        } else
            k::launch_spmv_nr(a, NR, x, y, dotOut, scal, it, ctl, nullptr, a.nChunk, s);
    }
}

// z = M^-1 r with the two-level preconditioner for NR vectors; the restricted residual is summed over the ranks
void tl_precond_nr(mfh_ctx *c, DistLink &L, int NR, const double *r, double *z, double *scal, int it, const double *ctl) {
    const k::TLArgs ta = tl_args(c);
    auto &T = c->tl;
    c->tlRcN.reserve((size_t)T.m * NR);
    c->tlYcN.reserve((size_t)T.m * NR);
    k::launch_tl_restrict_nr(ta, NR, T.aggPtr.p, T.dofsByAgg.p, r, c->tlRcN.p, c->stream);
    L.allreduce(c->tlRcN.p, T.m * NR);
    k::launch_tl_gemv_nr(T.m, T.ldInv, NR, T.Ainv.p, c->tlRcN.p, c->tlYcN.p, c->stream);
    k::launch_tl_apply_nr(ta, NR, c->dDinv.p, r, c->tlYcN.p, z, scal, it, ctl, c->stream);
}

}   // namespace

// ---- the same plumbing for callers outside this file (the distributed multigrid levels, mfh_multigrid.cpp)
bool dist_active(const mfh_ctx *c) { return c->dist.comm && c->dist.comm->world > 1; }
int dist_rank(const mfh_ctx *c) { return c->dist.comm ? c->dist.comm->rank : 0; }
int dist_world(const mfh_ctx *c) { return c->dist.comm ? c->dist.comm->world : 1; }
// y = K x on the owned rows; x holds nCols block rows, its halo part is fetched from the owners first
void dist_apply(mfh_ctx *c, double *x, double *y, bool masked) {
    DistLink L(c);
    if (cluster_operator(c) && c->mfcDev.ifaceBuf.n < (size_t)std::max<int64_t>(c->mfc.nIface, 1) * c->bs())
        c->mfcDev.ifaceBuf.alloc((size_t)std::max<int64_t>(c->mfc.nIface, 1) * c->bs());
    apply_op_nr(c, L, 1, x, y, masked, nullptr, nullptr, 0, nullptr, 0, true);      // (only the levels of the multigrid preconditioner come through here)
}
// y = K x for NR interleaved vectors on an unpartitioned context (the linear level of the batched V-cycle, mfh_multigrid.cpp)
void batch_apply(mfh_ctx *c, int NR, double *x, double *y, bool masked) {
    DistLink L(c);
    if (L.active) throw Error(MFH_ERR_UNSUPPORTED, "batched operator inside the multigrid preconditioner of a row-partitioned context");
    if (cluster_operator(c) && c->mfcDev.ifaceBuf.n < (size_t)std::max<int64_t>(c->mfc.nIface, 1) * c->bs() * NR)
        c->mfcDev.ifaceBuf.alloc((size_t)std::max<int64_t>(c->mfc.nIface, 1) * c->bs() * NR);
    apply_op_nr(c, L, NR, x, y, masked, nullptr, nullptr, 0, nullptr, 0, true);
}
// the halo block rows of v (nCols x W doubles) <- the owners' values
void dist_halo(mfh_ctx *c, double *v, int W) {
    DistLink L(c);
    if (!L.active) return;
    c->dist.sendBufW = std::max(c->dist.sendBufW, W);
    L.halo_begin(v, W);
    L.halo_end();
}
void dist_allreduce(mfh_ctx *c, double *dev, int64_t n) {
    DistLink L(c);
    L.allreduce(dev, n);
}
// ---- exchanges of a PARTITIONED aggregate level of the multigrid hierarchy (mfh_multigrid.cpp, localize_aggregate_levels). Same plumbing as
// the nodal halo exchange: pack on the compute stream, transfer on the communication stream, the compute stream waits for it.
void dist_level_forward(mfh_ctx *c, mfh_ctx::AggLevel &A, double *v, int W) {
    DistLink L(c);
    if (!L.active || A.xPeers.empty()) return;
    auto &D = c->dist;
    const int np = (int)A.xPeers.size();
    const int64_t nSend = A.xSendPtr.back();
    A.xSendBuf.reserve((size_t)std::max<int64_t>(1, nSend) * W);
    k::launch_pack_rows(nSend, W, A.xSendIdx.p, v, A.xSendBuf.p, L.s);
    MFH_HIP(hipEventRecord(D.ev[0], L.s));
    MFH_HIP(hipStreamWaitEvent(L.cs, D.ev[0], 0));
    std::vector<const double *> sb((size_t)np);
    std::vector<double *> rb((size_t)np);
    std::vector<int64_t> sc((size_t)np), rc((size_t)np);
    for (int k = 0; k < np; ++k) {
        sb[k] = A.xSendBuf.p + A.xSendPtr[k] * W;
        sc[k] = (A.xSendPtr[k + 1] - A.xSendPtr[k]) * W;
        rb[k] = v + (A.nOwn + A.xRecvPtr[k]) * W;
        rc[k] = (A.xRecvPtr[k + 1] - A.xRecvPtr[k]) * W;
    }
    comm_exchange(L.cm, np, A.xPeers.data(), sb.data(), sc.data(), rb.data(), rc.data(), L.cs);
    MFH_HIP(hipEventRecord(D.ev[1], L.cs));
    MFH_HIP(hipStreamWaitEvent(L.s, D.ev[1], 0));
}
void dist_level_reverse_add(mfh_ctx *c, mfh_ctx::AggLevel &A, double *v, int W) {
    DistLink L(c);
    if (!L.active || A.xPeers.empty()) return;
    auto &D = c->dist;
    const int np = (int)A.xPeers.size();
    const int64_t nSend = A.xSendPtr.back();
    A.xRecvBuf.reserve((size_t)std::max<int64_t>(1, nSend) * W);
    MFH_HIP(hipEventRecord(D.ev[0], L.s));
    MFH_HIP(hipStreamWaitEvent(L.cs, D.ev[0], 0));
    std::vector<const double *> sb((size_t)np);
    std::vector<double *> rb((size_t)np);
    std::vector<int64_t> sc((size_t)np), rc((size_t)np);
    for (int k = 0; k < np; ++k) {           // the halo entries of peer k lie together: sent where they are
        sb[k] = v + (A.nOwn + A.xRecvPtr[k]) * W;
        sc[k] = (A.xRecvPtr[k + 1] - A.xRecvPtr[k]) * W;
        rb[k] = A.xRecvBuf.p + A.xSendPtr[k] * W;
        rc[k] = (A.xSendPtr[k + 1] - A.xSendPtr[k]) * W;
    }
    comm_exchange(L.cm, np, A.xPeers.data(), sb.data(), sc.data(), rb.data(), rc.data(), L.cs);
    MFH_HIP(hipEventRecord(D.ev[1], L.cs));
    MFH_HIP(hipStreamWaitEvent(L.s, D.ev[1], 0));
    for (int k = 0; k < np; ++k)             // one peer after the other: an owned aggregate may be on several lists
        k::launch_unpack_add_rows(A.xSendPtr[k + 1] - A.xSendPtr[k], W, A.xSendIdx.p + A.xSendPtr[k], A.xRecvBuf.p + A.xSendPtr[k] * W, v, L.s);
}
// Setup helper: rank r hands every rank q the list toRank[q] and gets fromRank[q] back (lengths agreed through an all-reduce of the
// world x world count matrix, the lists travel as doubles through the communicator's exchange). Blocking.
void dist_exchange_lists(mfh_ctx *c, const std::vector<std::vector<int32_t>> &toRank, std::vector<std::vector<int32_t>> &fromRank) {
    DistLink L(c);
    const int world = dist_world(c), me = dist_rank(c);
    fromRank.assign((size_t)world, {});
    if (!L.active) return;
    hipStream_t s = L.s;
    std::vector<double> cnt((size_t)world * world, 0.0);
    for (int q = 0; q < world; ++q) cnt[(size_t)me * world + q] = (double)toRank[(size_t)q].size();
    DBuf<double> dCnt;
    dCnt.upload(cnt, s);
    L.allreduce(dCnt.p, (int64_t)cnt.size());
    dCnt.download(cnt.data(), cnt.size(), s);
    std::vector<int32_t> peers;
    std::vector<int64_t> sPtr{0}, rPtr{0};
    std::vector<double> hs;
    for (int q = 0; q < world; ++q) {
        if (q == me) continue;
        const int64_t ns = (int64_t)toRank[(size_t)q].size(), nr = (int64_t)cnt[(size_t)q * world + me];
        if (ns == 0 && nr == 0) continue;
        peers.push_back(q);
        for (int32_t v : toRank[(size_t)q]) hs.push_back((double)v);
        sPtr.push_back(sPtr.back() + ns);
        rPtr.push_back(rPtr.back() + nr);
    }
    if (peers.empty()) return;
    DBuf<double> dS, dR;
    dS.upload(hs.empty() ? std::vector<double>{0.0} : hs, s);
    dR.alloc((size_t)std::max<int64_t>(1, rPtr.back()));
    const int np = (int)peers.size();
    std::vector<const double *> sb((size_t)np);
    std::vector<double *> rb((size_t)np);
    std::vector<int64_t> sc((size_t)np), rc((size_t)np);
    for (int k = 0; k < np; ++k) { sb[k] = dS.p + sPtr[k]; sc[k] = sPtr[k + 1] - sPtr[k]; rb[k] = dR.p + rPtr[k]; rc[k] = rPtr[k + 1] - rPtr[k]; }
    MFH_HIP(hipStreamSynchronize(s));
    base_exchange(L.cm, np, peers.data(), sb.data(), sc.data(), rb.data(), rc.data(), s);      // (setup: the transport underneath, whatever the peer transfers are sized for)
    std::vector<double> hr((size_t)std::max<int64_t>(1, rPtr.back()));
    dR.download(hr.data(), hr.size(), s);
    for (int k = 0; k < np; ++k) {
        auto &dst = fromRank[(size_t)peers[k]];
        dst.resize((size_t)(rPtr[k + 1] - rPtr[k]));
        for (int64_t i = rPtr[k]; i < rPtr[k + 1]; ++i) dst[(size_t)(i - rPtr[k])] = (int32_t)hr[(size_t)i];
    }
}
// a second context on the same communicator whose nodes are a subset of c's (the linear level of the multigrid hierarchy): its
// exchange lists are c's, filtered. keep[n] >= 0: the node's id in the child (children numbered in c's node order).
void dist_setup_child(mfh_ctx *c, mfh_ctx *child, const std::vector<int32_t> &keep) {
    auto &D = c->dist;
    auto &E = child->dist;
    const int64_t nOwned = c->nOwnedDoF();              // block rows: nodes, or DoFs under a partitioned DoF map
    E.peers = D.peers;
    E.sendPtr.assign(1, 0); E.recvPtr.assign(1, 0);
    std::vector<int32_t> idx;
    for (size_t k = 0; k < D.peers.size(); ++k) {
        for (int64_t q = D.sendPtr[k]; q < D.sendPtr[k + 1]; ++q) {
            const int32_t ch = keep[(size_t)D.sendNodesHost[(size_t)q]];
            if (ch >= 0) idx.push_back(ch);
        }
        E.sendPtr.push_back((int64_t)idx.size());
        int64_t cnt = 0;
        for (int64_t q = D.recvPtr[k]; q < D.recvPtr[k + 1]; ++q) cnt += keep[(size_t)(nOwned + q)] >= 0;
        E.recvPtr.push_back(E.recvPtr.back() + cnt);
    }
    if (E.recvPtr.back() != child->nDoF - child->nOwnedDoF()) throw Error(MFH_ERR_STATE, "child halo does not match the filtered receive lists");
    E.sendNodesHost = idx;
    E.sendIdx.upload(idx.empty() ? std::vector<int32_t>{0} : idx, child->stream);
    // ONE communication stream per communicator: the child's collectives are issued on the parent's, in program order with them
    E.commStream = D.commStream;
    E.commStreamBorrowed = true;
    if (!E.ev[0])
        for (auto &e : E.ev) MFH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    mfh_comm *cm = D.comm;
    if (E.comm != cm) {
        dist_detach(child);
        if (cm) { std::lock_guard<std::mutex> lock(cm->mu); cm->users.push_back(child); }
    }
    E.comm = cm;
    E.listKind = 0;
    E.sendBufW = std::max(E.sendBufW, child->bs());
}

// averages of the timed events of option "dist_profile" (the stream is idle: the solve has downloaded its result)
void dist_profile_collect(mfh_ctx *c) {
    auto &D = c->dist;
    D.profiled = 0;
    D.exchangeMs = D.interiorMs = D.boundaryMs = D.exposedMs = D.operatorMs = 0;
    if (!D.profile || D.profUsed == 0) return;
    MFH_HIP(hipStreamSynchronize(c->stream));
    if (D.commStream) MFH_HIP(hipStreamSynchronize(D.commStream));
    // the first applications of a solve include one-time work (buffers, lists): skip two when there are enough
    const int first = D.profUsed > 6 ? 2 : 0;
    for (int q = first; q < D.profUsed; ++q) {
        auto &P = D.prof[(size_t)q];
        float ex = 0, t2 = 0, t3 = 0, t4 = 0;
        if (hipEventElapsedTime(&ex, P.ev[1], P.ev[2]) != hipSuccess || hipEventElapsedTime(&t2, P.ev[0], P.ev[2]) != hipSuccess ||
            hipEventElapsedTime(&t3, P.ev[0], P.ev[3]) != hipSuccess || hipEventElapsedTime(&t4, P.ev[0], P.ev[4]) != hipSuccess) { (void)hipGetLastError(); continue; }
        D.exchangeMs += ex; D.interiorMs += t3; D.boundaryMs += t4 - std::max(t2, t3); D.exposedMs += std::max(0.0f, t2 - t3); D.operatorMs += t4;
        ++D.profiled;
    }
    if (D.profiled) { const double f = 1.0 / D.profiled; D.exchangeMs *= f; D.interiorMs *= f; D.boundaryMs *= f; D.exposedMs *= f; D.operatorMs *= f; }
    D.profUsed = 0;
}

bool cg_operator_supported(mfh_ctx *c) {
    // the batched operators: the cluster variant of the matrix-free operator and the assembled SpMV
    if (!c->use_mf()) return true;
    prepare_matrix_free(c);
    return cluster_operator(c);
}

// Ranks agree on what the solve will do (see mfh_dist_solve). Collective: every rank of the communicator calls it.
void dist_agree(mfh_ctx *c) {
    DistLink L(c);
    std::string localErr;
    mfh_status localCode = MFH_OK;
    bool supported = false;
    auto prepare = [&]() {
        try {
            ensure_precond(c);
            supported = cg_operator_supported(c);
        } catch (const Error &e) { localErr = e.what(); localCode = e.code; }
    };
    prepare();
    for (int round = 0; round < 2; ++round) {
        const bool useTL = (c->precond == MFH_PRECOND_TWO_LEVEL || c->precond == MFH_PRECOND_MULTIGRID) && c->tl.valid;
        double h[8] = {c->anyFixedNonzero ? 1.0 : 0.0, localCode != MFH_OK ? 1.0 : 0.0, (localCode == MFH_OK && !supported) ? 1.0 : 0.0,
                       (localCode == MFH_OK && supported && cluster_operator(c)) ? 1.0 : 0.0, useTL ? 1.0 : 0.0, 1.0,
                       c->precond == MFH_PRECOND_MULTIGRID ? 1.0 : 0.0, 0.0};
        c->cgCtl.alloc(16);
        MFH_HIP(hipMemcpyAsync(c->cgCtl.p, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
        L.allreduce(c->cgCtl.p, 8);
        MFH_HIP(hipMemcpyAsync(h, c->cgCtl.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        MFH_HIP(hipStreamSynchronize(c->stream));
        const double world = h[5];
        c->dist.anyFixedNonzeroGlobal = h[0] > 0;
        c->dist.multigridAgreed = h[6] > 0 && h[6] == world;      // building the hierarchy is collective: every rank or none
        if (h[6] > 0 && h[6] < world) c->precondNote += " [multigrid requested on some ranks only: not used]";
        if (h[1] > 0) {
            if (localCode != MFH_OK) throw Error(localCode, localErr);
            throw Error(MFH_ERR_STATE, "another rank failed while preparing the distributed solve");
        }
        const bool mixedOperator = h[3] > 0 && h[3] < world;
        if (h[2] > 0 || mixedOperator) {
            // some rank cannot run the cluster operator (element order without locality there): all ranks move to the assembled
            // SpMV on both triangles of K, then agree again
            if (round == 1 || c->external) throw Error(MFH_ERR_UNSUPPORTED, "the distributed solve needs the cluster matrix-free operator or the assembled SpMV on every rank");
            c->matrixFree = 0;
            refresh_storage_rule(c);
            c->precondNote += " [a rank could not run the cluster operator: every rank uses the assembled SpMV]";
            prepare();
            continue;
        }
        if (h[4] > 0 && h[4] < world) {   // the coarse level is missing on some rank: block-Jacobi everywhere
            c->tl.valid = false;
            c->precondNote += " [two-level setup missing on a rank: block-Jacobi on every rank]";
        }
        return;
    }
}

// PCG on the free variables for NR right-hand sides at once. f / u: NR host vectors of bs * nRows doubles each (nRows = the
// rows this context owns), fStride doubles apart.
void solve_cg(mfh_ctx *c, int NR, const double *f, double *u, int64_t fStride, double rtol, int maxit, mfh_solve_info *infos) {
    RoctxRange range("Elasticity Solve");
    const int d = c->bs();
    const int W = NR * d;
    const int64_t nRows = c->sym.nRows, nCols = c->sym.nCols;
    const int64_t nOwn = nRows * W, nAll = nCols * W;
    hipStream_t s = c->stream;
    DistLink L(c);
    if (nRows != nCols && !c->dist.comm) throw Error(MFH_ERR_STATE, "row-partitioned context: call mfh_dist_setup before solving");
    if (!k::op_batch_supported(d, NR)) throw Error(MFH_ERR_UNSUPPORTED, "unsupported batch size");
    EventTimer tsetup(s);
    c->cgU.reserve(nAll); c->cgW.reserve(nOwn); c->cgP.reserve(nOwn); c->cgS.reserve(nOwn); c->cgX.reserve(nOwn); c->cgR.reserve(nOwn); c->cgF.reserve(nOwn);
    c->cgCtl.alloc(16);
    const bool masked = !c->fixedVars.empty();
    const bool cluster = cluster_operator(c);
    if (cluster && c->mfcDev.ifaceBuf.n < (size_t)std::max<int64_t>(c->mfc.nIface, 1) * W)
        c->mfcDev.ifaceBuf.alloc((size_t)std::max<int64_t>(c->mfc.nIface, 1) * W);
    if (L.active) c->dist.sendBufW = std::max(c->dist.sendBufW, W);
    // right-hand sides: NR separate host vectors -> interleaved on the device
    for (int k2 = 0; k2 < NR; ++k2)
        MFH_HIP(hipMemcpyAsync(c->cgS.p + (size_t)k2 * nRows * d, f + (size_t)k2 * fStride, (size_t)nRows * d * sizeof(double), hipMemcpyHostToDevice, s));
    k::launch_interleave(nRows, NR, d, c->cgS.p, c->cgF.p, true, nRows * d, s);
    // b = f - K ubar on the free variables (SparseMatrices.hh:2457-2470,2526-2535); b lives in R
    MFH_HIP(hipMemcpyAsync(c->cgR.p, c->cgF.p, (size_t)nOwn * sizeof(double), hipMemcpyDeviceToDevice, s));
    // the lift exchanges halo entries: with a communicator EVERY rank applies it (a non-zero value may be visible to some ranks only)
    if ((c->anyFixedNonzero || (L.active && c->dist.anyFixedNonzeroGlobal)) && !c->solveHomogeneous) {
        c->cgU.zero(s);
        k::launch_scatter_values_nr((int64_t)c->fixedVars.size(), NR, d, c->dFixedIdx.p, c->dFixedVal.p, c->cgU.p, nCols, s);
        apply_op_nr(c, L, NR, c->cgU.p, c->cgW.p, false, nullptr, nullptr, 0, nullptr);
        k::launch_axpby(nOwn, -1.0, c->cgW.p, 1.0, c->cgR.p, s);
    }
    if (masked) k::launch_mask_nr(nRows, NR, d, c->dFixedMask.p, c->cgR.p, s);
    double bb[8] = {0};
    {
        MFH_HIP(hipMemsetAsync(c->cgCtl.p, 0, 16 * sizeof(double), s));
        k::launch_norms_nr(nRows, NR, d, c->cgR.p, c->cgCtl.p + 8, s);
        L.allreduce(c->cgCtl.p + 8, NR);
        MFH_HIP(hipMemcpyAsync(bb, c->cgCtl.p + 8, NR * sizeof(double), hipMemcpyDeviceToHost, s));
        MFH_HIP(hipStreamSynchronize(s));
    }
    double ctlHost[16] = {0};
    bool anyWork = false;
    for (int k2 = 0; k2 < NR; ++k2) { ctlHost[2 + k2] = rtol * rtol * bb[k2]; anyWork |= bb[k2] > 0; }
    MFH_HIP(hipMemcpyAsync(c->cgCtl.p, ctlHost, 16 * sizeof(double), hipMemcpyHostToDevice, s));
    const size_t scalN = ((size_t)maxit + (size_t)c->checkEvery + 3) * 4 * NR;
    c->scal.alloc(scalN);
    c->scal.zero(s);
    c->cgX.zero(s); c->cgP.zero(s); c->cgS.zero(s);
    const double setupMs = tsetup.stop();
    std::vector<int> itConv((size_t)NR, -1);
    std::vector<double> rrFinal((size_t)NR, 0.0);
    double solveMs = 0;
    bool usedGraph = false;
    int itRun = 0;
    if (anyWork) {
        EventTimer tsolve(s);
        const bool useTL = (c->precond == MFH_PRECOND_TWO_LEVEL || c->precond == MFH_PRECOND_MULTIGRID) && c->tl.valid && !c->tlSuppress;
        double *scal = c->scal.p;
        const double *ctl = c->cgCtl.p;
        // start: u = M^-1 r, w = K u, {gamma, delta, rr}_0
        k::launch_cg_init(d, nRows, NR, c->dDinv.p, c->cgR.p, c->cgU.p, scal, useTL, s);
        if (useTL) tl_precond_nr(c, L, NR, c->cgR.p, c->cgU.p, scal, -1, ctl);
        apply_op_nr(c, L, NR, c->cgU.p, c->cgW.p, masked, scal + 1, nullptr, 0, nullptr);
        L.allreduce(scal, 4 * NR);
        auto enqueue = [&](int itLocal, int itAbs) {
            k::launch_cg_update(d, nRows, NR, c->dDinv.p, c->cgU.p, c->cgW.p, c->cgP.p, c->cgS.p, c->cgX.p, c->cgR.p, scal, itLocal, ctl, useTL, s);
            if (useTL) tl_precond_nr(c, L, NR, c->cgR.p, c->cgU.p, scal, itLocal, ctl);
            apply_op_nr(c, L, NR, c->cgU.p, c->cgW.p, masked, nullptr, scal, itLocal, ctl);
            L.allreduce(scal + (size_t)(itAbs + 1) * 4 * NR, 4 * NR);
        };
        // launch-bound regime (small meshes): replay blocks of check_every iterations from a hipGraph; the kernels find their
        // iteration through the device-side base ctl[0], which the last node advances. Not with a communicator.
        hipGraphExec_t exec = nullptr;
        if (c->useGraph && c->checkEvery > 1 && !L.active) {
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
                bool ok = true;
                try {
                    for (int j = 0; j < c->checkEvery; ++j) enqueue(j, j);
                    k::launch_add_scalar(c->cgCtl.p, (double)c->checkEvery, s);
                } catch (...) { ok = false; }
                if (hipStreamEndCapture(s, &graph) != hipSuccess || !ok || !graph) { graph = nullptr; (void)hipGetLastError(); }
            } else (void)hipGetLastError();
            if (graph) {
                if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { exec = nullptr; (void)hipGetLastError(); }
                (void)hipGraphDestroy(graph);
            }
        }
        usedGraph = exec != nullptr;
        std::vector<double> hs;
        int it = 0, lastChecked = 0, nConv = 0;
        for (int k2 = 0; k2 < NR; ++k2) if (!(bb[k2] > 0)) { itConv[k2] = 0; ++nConv; }
        std::vector<double> bestRR((size_t)NR, 1e300);
        std::vector<int> itBest((size_t)NR, 0);
        // (a plateau is not a stagnation: block-Jacobi PCG on a one-layer plate in bending, 59 k DOF, sits above its best residual for more than
        // 5 000 iterations and then converges at 5 913 -- CG owes its answer within about n iterations, so the window grows with n)
        const int stagnationWindow = (int)std::max<int64_t>(std::max(5000, 40 * c->checkEvery), std::min<int64_t>((int64_t)d * c->nDoF, 50000));
        while (nConv < NR && it < maxit) {
            if (exec) { MFH_HIP(hipGraphLaunch(exec, s)); it += c->checkEvery; }
            else {
                const int itEnd = std::min(maxit, it + c->checkEvery);
                for (; it < itEnd; ++it) enqueue(it, it);
            }
            hs.resize((size_t)(it - lastChecked + 1) * 4 * NR);
            MFH_HIP(hipMemcpyAsync(hs.data(), scal + (size_t)lastChecked * 4 * NR, hs.size() * sizeof(double), hipMemcpyDeviceToHost, s));
            MFH_HIP(hipStreamSynchronize(s));
            for (int k2 = 0; k2 < NR; ++k2) {
                if (itConv[k2] >= 0) continue;
                for (int q = lastChecked; q <= std::min(it, maxit); ++q) {
                    const double *sc = &hs[((size_t)(q - lastChecked) * NR + k2) * 4];
                    const double rr = sc[2];
                    if (rr <= ctlHost[2 + k2]) { itConv[k2] = q; rrFinal[k2] = rr; ++nConv; break; }
                    if (!(rr == rr)) throw Error(MFH_ERR_NOT_CONVERGED, "PCG breakdown (NaN residual): K is not SPD on the free variables");
                    // CHOLMOD reports "not positive definite" at once; the iterative counterpart: negative curvature
                    // (p.Kp = gamma / alpha < 0), or a residual that has not improved by 10 % for thousands of iterations (a
                    // singular system with an inconsistent right-hand side: missing boundary conditions, unbalanced loads)
                    if (q < it && sc[3] != 0.0 && sc[0] / sc[3] < 0.0)
                        throw Error(MFH_ERR_NOT_CONVERGED, "PCG breakdown (p.Kp = " + std::to_string(sc[0] / sc[3]) + " < 0 at iteration " + std::to_string(q) +
                                                               ", residual^2 " + std::to_string(rr) + "): K is not positive definite on the free variables");
                    if (rr < 0.9 * bestRR[k2]) { bestRR[k2] = rr; itBest[k2] = q; }
                    else if (q - itBest[k2] > stagnationWindow)
                        throw Error(MFH_ERR_NOT_CONVERGED, "PCG stagnated (no progress of the residual for " + std::to_string(stagnationWindow) +
                                                               " iterations): the system is singular with an inconsistent right-hand side "
                                                               "(missing boundary conditions?) or too ill-conditioned for this preconditioner");
                    rrFinal[k2] = rr;
                }
            }
            lastChecked = it;
        }
        itRun = it;
        if (exec) (void)hipGraphExecDestroy(exec);
        solveMs = tsolve.stop();
    } else {
        for (int k2 = 0; k2 < NR; ++k2) itConv[k2] = 0;
    }
    // u = x + ubar  (SparseMatrices.hh:2592-2605)
    if (masked && !c->solveHomogeneous) k::launch_scatter_values_nr((int64_t)c->fixedVars.size(), NR, d, c->dFixedIdx.p, c->dFixedVal.p, c->cgX.p, nRows, s);   // owned rows only: x has no halo part
    // true residual on the free variables: || mask(f - K u) || / ||b||
    double tr[8] = {0};
    if (anyWork) {
        MFH_HIP(hipMemsetAsync(c->cgCtl.p, 0, 16 * sizeof(double), s));   // the operator below must not be gated
        MFH_HIP(hipMemcpyAsync(c->cgU.p, c->cgX.p, (size_t)nOwn * sizeof(double), hipMemcpyDeviceToDevice, s));
        if (nAll > nOwn && !L.active) MFH_HIP(hipMemsetAsync(c->cgU.p + nOwn, 0, (size_t)(nAll - nOwn) * sizeof(double), s));
        if (nAll > nOwn && masked && !c->solveHomogeneous)   // fixed values on halo variables (also refreshed by the exchange)
            k::launch_scatter_values_nr((int64_t)c->fixedVars.size(), NR, d, c->dFixedIdx.p, c->dFixedVal.p, c->cgU.p, nCols, s);
        apply_op_nr(c, L, NR, c->cgU.p, c->cgW.p, false, nullptr, nullptr, 0, nullptr);
        k::launch_axpby(nOwn, 1.0, c->cgF.p, -1.0, c->cgW.p, s);
        if (masked) k::launch_mask_nr(nRows, NR, d, c->dFixedMask.p, c->cgW.p, s);
        k::launch_norms_nr(nRows, NR, d, c->cgW.p, c->cgCtl.p + 8, s);
        L.allreduce(c->cgCtl.p + 8, NR);
        MFH_HIP(hipMemcpyAsync(tr, c->cgCtl.p + 8, NR * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    k::launch_interleave(nRows, NR, d, c->cgX.p, c->cgS.p, false, nRows * d, s);
    for (int k2 = 0; k2 < NR; ++k2)
        MFH_HIP(hipMemcpyAsync(u + (size_t)k2 * fStride, c->cgS.p + (size_t)k2 * nRows * d, (size_t)nRows * d * sizeof(double), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    mfh_solve_info gap[8];
    for (int k2 = 0; k2 < NR; ++k2) {
        mfh_solve_info li{};
        li.converged = itConv[k2] >= 0 ? 1 : 0;
        li.iterations = itConv[k2] >= 0 ? itConv[k2] : std::min(itRun, maxit);
        li.rel_residual = bb[k2] > 0 ? std::sqrt(rrFinal[k2] / bb[k2]) : 0.0;
        li.true_rel_residual = bb[k2] > 0 ? std::sqrt(tr[k2] / bb[k2]) : 0.0;
        li.solve_ms = solveMs;
        li.setup_ms = setupMs;
        li.used_graph = usedGraph ? 1 : 0;
        li.reserved = NR;
        if (infos) infos[k2] = li;
        gap[k2] = li;
    }
    for (int k2 = 0; k2 < NR; ++k2) check_residual_gap(gap[k2], rtol);   // after every info has been written (norms are global: all ranks agree)
}

// The CLASSIC PCG (two reduction points) on a row-partitioned context, one right-hand side: the same kernels as the
// single-GPU classic loop (one vector pass less per iteration than Chronopoulos-Gear, 0.945 vs 1.089 ms at config 3), at the
// price of a second all-reduce per iteration (p.Ap after the operator; {r.z, r.r} after the preconditioner). Which loop is
// faster on N GPUs depends on the node's all-reduce latency: option "dist_pcg_variant" selects, bench.py measures both.
void solve_classic_partitioned(mfh_ctx *c, const double *f, double *u, double rtol, int maxit, mfh_solve_info *info) {
    RoctxRange range("Elasticity Solve");
    const int d = c->bs();
    const int64_t nRows = c->sym.nRows, nCols = c->sym.nCols, nOwn = nRows * d, nAll = nCols * d;
    hipStream_t s = c->stream;
    DistLink L(c);
    EventTimer tsetup(s);
    c->cgU.reserve(nAll);                                 // p (with its halo part)
    // the multigrid V-cycle applies the operator to z: it then carries its halo rows like p
    const bool useMG = c->precond == MFH_PRECOND_MULTIGRID && c->mg.valid && c->mg.distributed == L.active && c->mg.singular == c->tlSuppress;
    c->wx.alloc(nOwn); c->wr.alloc(nOwn); c->wz.alloc(useMG ? nAll : nOwn); c->wAp.alloc(nOwn); c->wf.alloc(nOwn);
    if (useMG) c->wz.zero(s);
    c->stop.alloc(4);
    c->cgCtl.alloc(16);
    const bool masked = !c->fixedVars.empty();
    if (cluster_operator(c) && c->mfcDev.ifaceBuf.n < (size_t)std::max<int64_t>(c->mfc.nIface, 1) * d) c->mfcDev.ifaceBuf.alloc((size_t)std::max<int64_t>(c->mfc.nIface, 1) * d);
    if (L.active) c->dist.sendBufW = std::max(c->dist.sendBufW, d);
    double *p = c->cgU.p;
    MFH_HIP(hipMemcpyAsync(c->wf.p, f, (size_t)nOwn * sizeof(double), hipMemcpyHostToDevice, s));
    MFH_HIP(hipMemcpyAsync(c->wr.p, c->wf.p, (size_t)nOwn * sizeof(double), hipMemcpyDeviceToDevice, s));
    if ((c->anyFixedNonzero || (L.active && c->dist.anyFixedNonzeroGlobal)) && !c->solveHomogeneous) {     // b = f - K ubar (SparseMatrices.hh:2457-2470,2526-2535); decided globally
        c->cgU.zero(s);
        k::launch_scatter_values((int64_t)c->fixedVars.size(), c->dFixedIdx.p, c->dFixedVal.p, p, nAll, s);
        apply_op_nr(c, L, 1, p, c->wAp.p, false, nullptr, nullptr, 0, nullptr);
        k::launch_axpby(nOwn, -1.0, c->wAp.p, 1.0, c->wr.p, s);
    }
    if (masked) k::launch_mask(nOwn, c->dFixedMask.p, c->wr.p, s);
    double bb = 0;
    MFH_HIP(hipMemsetAsync(c->cgCtl.p, 0, 16 * sizeof(double), s));
    k::launch_dot(nOwn, c->wr.p, c->wr.p, c->cgCtl.p + 8, s);
    L.allreduce(c->cgCtl.p + 8, 1);
    MFH_HIP(hipMemcpyAsync(&bb, c->cgCtl.p + 8, sizeof(double), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    const size_t scalN = ((size_t)maxit + (size_t)c->checkEvery + 2) * 4;
    c->scal.alloc(scalN);
    c->scal.zero(s);
    const double stopv = rtol * rtol * bb;
    MFH_HIP(hipMemsetAsync(c->stop.p, 0, 4 * sizeof(double), s));
    MFH_HIP(hipMemcpyAsync(c->stop.p, &stopv, sizeof(double), hipMemcpyHostToDevice, s));
    mfh_solve_info li{};
    li.setup_ms = tsetup.stop();
    li.reserved = 1;
    int itDone = 0;
    double rrFinal = 0;
    if (bb == 0.0) { c->wx.zero(s); li.converged = 1; }
    else {
        EventTimer tsolve(s);
        const bool useTL = !useMG && (c->precond == MFH_PRECOND_TWO_LEVEL || c->precond == MFH_PRECOND_MULTIGRID) && c->tl.valid && !c->tlSuppress;
        const uint8_t *maskPtr = masked ? c->dFixedMask.p : nullptr;
        const int checkEvery = useMG ? std::min(c->checkEvery, 2) : c->checkEvery;     // a V-cycle is milliseconds long and tens of them are needed
        double *scal = c->scal.p;
        auto &T = c->tl;
        auto tl_pre = [&](int it) {     // z = M^-1 r, r.z into scal[(it + 1) 4]; the restricted residual summed over the ranks
            const k::TLArgs ta = tl_args(c);
            k::launch_tl_restrict(ta, T.aggPtr.p, T.dofsByAgg.p, c->wr.p, T.rc.p, s);
            L.allreduce(T.rc.p, T.m);
            k::launch_tl_gemv(T.m, T.ldInv, T.Ainv.p, T.rc.p, T.yc.p, s);
            k::launch_tl_apply(ta, c->dDinv.p, c->wr.p, T.yc.p, c->wz.p, scal, it, c->stop.p, s);
        };
        // x = 0, r = b, z = D^-1 r, p = z; {r.z, r.r}_0
        MFH_HIP(hipMemsetAsync(p, 0, (size_t)nAll * sizeof(double), s));
        MFH_HIP(hipMemcpyAsync(c->wAp.p, c->wr.p, (size_t)nOwn * sizeof(double), hipMemcpyDeviceToDevice, s));
        k::launch_pcg_init(d, nRows, c->dDinv.p, c->wAp.p, c->wx.p, c->wr.p, c->wz.p, p, scal, s);
        if (useTL || useMG) {
            MFH_HIP(hipMemsetAsync(scal, 0, sizeof(double), s));
            if (useMG) {      // z = M^-1 r by one V-cycle (halo exchanges and one small all-reduce inside), r.z into scal[(it + 1) 4]
                mg_precond(c, c->wr.p, c->wz.p, nullptr, -1, nullptr);
                k::launch_mg_rz(nOwn, c->wr.p, c->wz.p, maskPtr, scal, -1, nullptr, nullptr, s);
            } else
                tl_pre(-1);
            MFH_HIP(hipMemcpyAsync(p, c->wz.p, (size_t)nOwn * sizeof(double), hipMemcpyDeviceToDevice, s));
        }
        L.allreduce(scal, 3);
        std::vector<double> hs;
        int it = 0, lastChecked = 0;
        bool done = false;
        while (!done && it < maxit) {
            const int itEnd = std::min(maxit, it + checkEvery);
            for (; it < itEnd; ++it) {
                apply_op_nr(c, L, 1, p, c->wAp.p, masked, nullptr, scal, it, c->stop.p, 1);     // Ap, p.Ap (first reduction point)
                L.allreduce(scal + (size_t)it * 4 + 1, 1);
                if (useMG) {
                    k::launch_pcg_update_noz(d, nRows, c->wAp.p, c->wr.p, scal, it, c->stop.p, s);
                    mg_precond(c, c->wr.p, c->wz.p, scal, it, c->stop.p);
                    k::launch_mg_rz(nOwn, c->wr.p, c->wz.p, maskPtr, scal, it, scal, c->stop.p, s);
                } else if (useTL) { k::launch_pcg_update_noz(d, nRows, c->wAp.p, c->wr.p, scal, it, c->stop.p, s); tl_pre(it); }
                else k::launch_pcg_update(d, nRows, c->dDinv.p, c->wAp.p, c->wr.p, c->wz.p, scal, it, c->stop.p, s);
                L.allreduce(scal + (size_t)(it + 1) * 4, 3);                                    // {r.z, -, r.r} (second reduction point)
                k::launch_pcg_direction(nOwn, c->wz.p, p, c->wx.p, scal, it, c->stop.p, s);
            }
            hs.resize((size_t)(it - lastChecked + 1) * 4);
            MFH_HIP(hipMemcpyAsync(hs.data(), scal + (size_t)lastChecked * 4, hs.size() * sizeof(double), hipMemcpyDeviceToHost, s));
            MFH_HIP(hipStreamSynchronize(s));
            for (int k2 = lastChecked; k2 <= it; ++k2) {
                const double rr = hs[(size_t)(k2 - lastChecked) * 4 + 2];
                if (rr <= stopv) { done = true; itDone = k2; rrFinal = rr; break; }
                if (!(rr == rr)) throw Error(MFH_ERR_NOT_CONVERGED, "PCG breakdown (NaN residual): K is not SPD on the free variables");
                const double pAp = hs[(size_t)(k2 - lastChecked) * 4 + 1];
                if (k2 < it && pAp < 0.0) throw Error(MFH_ERR_NOT_CONVERGED, "PCG breakdown (p.Kp < 0): K is not positive definite on the free variables");
            }
            if (!done) { itDone = it; rrFinal = hs[(size_t)(it - lastChecked) * 4 + 2]; }
            lastChecked = it;
        }
        li.solve_ms = tsolve.stop();
        li.converged = done ? 1 : 0;
    }
    li.iterations = itDone;
    li.rel_residual = bb > 0 ? std::sqrt(rrFinal / bb) : 0.0;
    if (masked && !c->solveHomogeneous) k::launch_scatter_values((int64_t)c->fixedVars.size(), c->dFixedIdx.p, c->dFixedVal.p, c->wx.p, nOwn, s);   // owned rows only
    if (bb > 0) {   // true residual on the free variables
        MFH_HIP(hipMemcpyAsync(p, c->wx.p, (size_t)nOwn * sizeof(double), hipMemcpyDeviceToDevice, s));
        if (nAll > nOwn && masked && !c->solveHomogeneous) k::launch_scatter_values((int64_t)c->fixedVars.size(), c->dFixedIdx.p, c->dFixedVal.p, p, nAll, s);
        apply_op_nr(c, L, 1, p, c->wAp.p, false, nullptr, nullptr, 0, nullptr);
        k::launch_axpby(nOwn, 1.0, c->wf.p, -1.0, c->wAp.p, s);
        if (masked) k::launch_mask(nOwn, c->dFixedMask.p, c->wAp.p, s);
        MFH_HIP(hipMemsetAsync(c->cgCtl.p + 8, 0, sizeof(double), s));
        k::launch_dot(nOwn, c->wAp.p, c->wAp.p, c->cgCtl.p + 8, s);
        L.allreduce(c->cgCtl.p + 8, 1);
        double tr = 0;
        MFH_HIP(hipMemcpyAsync(&tr, c->cgCtl.p + 8, sizeof(double), hipMemcpyDeviceToHost, s));
        MFH_HIP(hipStreamSynchronize(s));
        li.true_rel_residual = std::sqrt(tr / bb);
    }
    MFH_HIP(hipMemcpyAsync(u, c->wx.p, (size_t)nOwn * sizeof(double), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    if (info) *info = li;
    check_residual_gap(li, rtol);
}

// The right-hand side of solve_one may already lie on the device, in c->wf (f == nullptr; neumannLoad formed by a kernel, mfh_simulator.cpp): the
// classic loop of an unpartitioned context takes it from there. (The Chronopoulos-Gear loop interleaves host vectors: not for it.)
bool device_rhs_supported(mfh_ctx *c) {
    return !c->hostOnly && c->sym.nRows == c->sym.nCols && !dist_active(c) && !c->deterministic && c->pcgVariant != 1;
}

// one right-hand side: the Chronopoulos-Gear loop when the operator in use has a batched kernel, else the classic PCG
static void solve_one_pass(mfh_ctx *c, const double *f, double *u, double rtol, int maxit, mfh_solve_info *info) {
    const bool partitioned = c->sym.nRows != c->sym.nCols;
    // the hierarchy for THIS solve (regular, or pinned for a system that is singular on the free variables): a no-op once it exists
    if (c->precond == MFH_PRECOND_MULTIGRID && !partitioned && !(c->mg.valid && c->mg.singular == c->tlSuppress)) ensure_coarse_levels(c, 1);
    const bool multigrid = c->precond == MFH_PRECOND_MULTIGRID && c->mg.valid && !partitioned;   // the V-cycle lives in the classic loop
    if (!f && !(!partitioned && c->pcgVariant != 1 && !c->deterministic)) throw Error(MFH_ERR_STATE, "device-resident right-hand side: classic loop only");
    if (!multigrid && !c->deterministic && (c->pcgVariant == 1 || partitioned) && cg_operator_supported(c)) {
        const int64_t n = (int64_t)c->bs() * c->sym.nRows;
        solve_cg(c, 1, f, u, n, rtol, maxit, info);
    } else {
        solve_one_classic(c, f, u, rtol, maxit, info);
        if (info) info->reserved = 1;
    }
}

// One right-hand side on a context that owns all its rows, with ITERATIVE REFINEMENT: over thousands of iterations the recurrence residual
// of the PCG drifts away from f - K u (block-Jacobi on a thin plate: 9.5e-9 reached, 3.9e-8 true), and the answer of the direct solver this
// replaces has no such gap. When the true residual ends above twice the tolerance, the correction K du = f - K u (du = 0 on the fixed
// variables) is solved to what is missing and added; option "refine" 0 switches it off. Solves that end within the tolerance -- all of the
// benchmark's -- never get here.
void solve_one(mfh_ctx *c, const double *f, double *u, double rtol, int maxit, mfh_solve_info *info) {
    mfh_solve_info li{};
    solve_one_pass(c, f, u, rtol, maxit, &li);
    const bool partitioned = c->sym.nRows != c->sym.nCols;
    if (c->refine && !partitioned && !c->solveHomogeneous && rtol > 0) {
        const int64_t n = (int64_t)c->bs() * c->nDoF;
        RawVec<double> r, du;
        DBuf<double> dU, dKu;
        for (int pass = 0; pass < 3 && li.converged && li.true_rel_residual > 2.0 * rtol && li.true_rel_residual < 1.0; ++pass) {
            if (!f) {        // a device-resident right-hand side (still in c->wf: the loop does not write it) comes to the host for the refinement
                resize_prefaulted(c->hLoad, (size_t)n);
                c->wf.download(c->hLoad.data(), (size_t)n, c->stream);
                f = c->hLoad.data();
            }
            r.resize((size_t)n); du.resize((size_t)n);
            dU.alloc((size_t)n); dKu.alloc((size_t)n);
            MFH_HIP(hipMemcpyAsync(dU.p, u, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
            apply_operator(c, false, dU.p, dKu.p, nullptr);
            dKu.download(r.data(), (size_t)n, c->stream);
            const uint8_t *mask = c->fixedVars.empty() ? nullptr : c->hFixedMask.data();
            parallel_ranges(n, [&](int64_t b, int64_t e, int) {
                for (int64_t i = b; i < e; ++i) r[(size_t)i] = (mask && mask[i]) ? 0.0 : f[i] - r[(size_t)i];
            });
            mfh_solve_info l2{};
            const double before = li.true_rel_residual;
            c->solveHomogeneous = true;
            try { solve_one_pass(c, r.data(), du.data(), std::min(0.5, rtol / before), maxit, &l2); }
            catch (...) { c->solveHomogeneous = false; throw; }
            c->solveHomogeneous = false;
            parallel_ranges(n, [&](int64_t b, int64_t e, int) {
                for (int64_t i = b; i < e; ++i) u[i] += du[(size_t)i];
            });
            li.iterations += l2.iterations;
            li.solve_ms += l2.solve_ms;
            li.setup_ms += l2.setup_ms;
            li.rel_residual = l2.rel_residual * before;        // both are relative to the residual the pass started from
            li.true_rel_residual = l2.true_rel_residual * before;
            li.converged = l2.converged;
        }
    }
    if (info) *info = li;
}

// Multigrid preconditioner, several right-hand sides, unpartitioned quadratic context (option "mg_batch", default on): one classic PCG loop per
// right-hand side on the quadratic level, advancing in lockstep, and ONE pass through the linear and aggregate levels of the V-cycle for
// all of them -- those levels are bound by their matrices and by launch latency, both shared by the batch (the reference factors once and
// back-substitutes per right-hand side, PeriodicHomogenization.hh:34-54, SparseMatrices.hh:2106-2124).
bool multigrid_batch_ready(mfh_ctx *c, int nrhs) {
    const bool partitioned = c->sym.nRows != c->sym.nCols;
    if (!(c->precond == MFH_PRECOND_MULTIGRID && c->mgBatch && nrhs > 1 && !partitioned && !c->deterministic && !dist_active(c))) return false;
    if (!(c->mg.valid && c->mg.singular == c->tlSuppress)) ensure_coarse_levels(c, nrhs);      // the hierarchy for THESE solves (see solve_one_pass)
    return c->mg.valid && !c->mg.distributed && c->mg.singular == c->tlSuppress && !c->mg.linearOnly && cg_operator_supported(c) && cluster_operator(c);
}

// nrhs right-hand sides in batches of the sizes the kernels are built for (3D: 6, 2, 1; 2D: 3, 1)
void solve_many(mfh_ctx *c, int nrhs, const double *f, double *u, int64_t stride, double rtol, int maxit, mfh_solve_info *infos) {
    const int d = c->bs();
    const bool mgBatch = multigrid_batch_ready(c, nrhs);
    // the Chronopoulos-Gear batches (option "batch_rhs") have no V-cycle: with the multigrid preconditioner they would silently run block-Jacobi
    const bool batched = c->pcgVariant != 0 && c->batchRhs && !c->deterministic && c->precond != MFH_PRECOND_MULTIGRID && cg_operator_supported(c);
    int k0 = 0;
    while (k0 < nrhs) {
        int nb = 1;
        if (batched || mgBatch)
            for (int cand : {6, 3, 2})
                if (cand <= nrhs - k0 && k::op_batch_supported(d, cand)) { nb = cand; break; }
        if (nb > 1 && mgBatch) {
            mfh_solve_info bi[6];
            solve_multigrid_batch(c, nb, f + (size_t)k0 * stride, u + (size_t)k0 * stride, stride, rtol, maxit, bi);
            // a right-hand side whose true residual ended above twice the tolerance gets what solve_one gives every single solve: iterative refinement
            for (int q = 0; q < nb; ++q) {
                if (c->refine && rtol > 0 && bi[q].converged && bi[q].true_rel_residual > 2.0 * rtol && bi[q].true_rel_residual < 1.0)
                    solve_one(c, f + (size_t)(k0 + q) * stride, u + (size_t)(k0 + q) * stride, rtol, maxit, &bi[q]);
                if (infos) infos[k0 + q] = bi[q];
            }
        } else if (nb > 1) solve_cg(c, nb, f + (size_t)k0 * stride, u + (size_t)k0 * stride, stride, rtol, maxit, infos ? infos + k0 : nullptr);
        else solve_one(c, f + (size_t)k0 * stride, u + (size_t)k0 * stride, rtol, maxit, infos ? infos + k0 : nullptr);
        k0 += nb;
    }
}

}   // namespace mfhi

// ------------------------------------------------------------------------------------------------ C ABI
using namespace mfhi;

extern "C" {

mfh_status mfh_rccl_get_unique_id(mfh_rccl_unique_id *id) {
    if (!id) return MFH_ERR_INVALID;
    RcclApi &a = rccl();
    if (!a.ok) return MFH_ERR_UNSUPPORTED;
    return a.getUniqueId(id) == 0 ? MFH_OK : MFH_ERR_HIP;
}

mfh_status mfh_comm_create_rccl(mfh_ctx *c, const mfh_rccl_unique_id *id, int32_t rank, int32_t world, mfh_comm **out) {
    MFH_TRY(c)
    require(c && id && out && world >= 1 && rank >= 0 && rank < world, MFH_ERR_INVALID, "bad communicator arguments");
    require_device(c);
    RcclApi &a = rccl();
    if (!a.ok) throw Error(MFH_ERR_UNSUPPORTED, "RCCL not found (tried the process, librccl.so.1, librccl.so; set MFH_RCCL_LIB)");
    MFH_HIP(hipSetDevice(c->device));
    std::unique_ptr<mfh_comm> cm(new mfh_comm());
    cm->rank = rank; cm->world = world; cm->device = c->device;
    rccl_check(a.commInitRank(&cm->nccl, world, *id, rank), "ncclCommInitRank");
    int ver = 0;
    if (a.getVersion) (void)a.getVersion(&ver);
    cm->desc = "RCCL " + std::to_string(ver) + " (" + a.where + "), " + std::to_string(world) + " ranks";
    *out = cm.release();
    MFH_CATCH(c)
}

mfh_status mfh_comm_create_callbacks(int32_t rank, int32_t world, void *user, mfh_allreduce_fn allreduce_sum, mfh_exchange_fn exchange,
                                     mfh_comm **out) {
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && (!allreduce_sum || !exchange))) return MFH_ERR_INVALID;
    mfh_comm *cm = new mfh_comm();
    cm->rank = rank; cm->world = world; cm->user = user; cm->allreduce = allreduce_sum; cm->exchange = exchange;
    cm->desc = "caller callbacks, " + std::to_string(world) + " ranks";
    *out = cm;
    return MFH_OK;
}

void mfh_comm_destroy(mfh_comm *cm) {
    if (!cm) return;
    for (mfh_ctx *c : cm->users) { c->dist.comm = nullptr; c->dist.listKind = 0; }   // no context keeps a dangling pointer
    cm->users.clear();
    peer_release(cm);
    if (cm->nccl && rccl().ok) (void)rccl().commDestroy(cm->nccl);
    delete cm;
}

const char *mfh_comm_describe(const mfh_comm *cm) { return cm ? ((cm->peer.enabled && !cm->descFull.empty()) ? cm->descFull.c_str() : cm->desc.c_str()) : ""; }

// Direct device-to-device transfers on top of an existing communicator (mfh_peer.hip). Collective; the IPC handles travel through the
// communicator's own all-reduce. Fails (and leaves the communicator as it was) when the ranks are not separate processes of one node.
mfh_status mfh_comm_enable_peer(mfh_ctx *c, mfh_comm *cm) {
    MFH_TRY(c)
    require(c && cm, MFH_ERR_INVALID, "bad arguments");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    try { peer_enable(cm, c->device, c->stream); }
    catch (const Error &e) { cm->peer.why = e.what(); throw; }
    MFH_CATCH(c)
}

mfh_status mfh_comm_disable_peer(mfh_ctx *c, mfh_comm *cm) {
    MFH_TRY(c)
    require(c && cm, MFH_ERR_INVALID, "bad arguments");
    if (cm->peer.enabled) {
        MFH_HIP(hipSetDevice(c->device));
        MFH_HIP(hipDeviceSynchronize());
        // the other ranks may still be reading this rank's flags: leave together
        double one = 1.0;
        DBuf<double> d;
        d.upload(&one, 1, c->stream);
        base_allreduce(cm, d.p, 1, c->stream);
        MFH_HIP(hipStreamSynchronize(c->stream));
        peer_release(cm);
    }
    MFH_CATCH(c)
}

mfh_status mfh_dist_get_stats(mfh_ctx *c, mfh_dist_stats *out) {
    MFH_TRY(c)
    require(c && out, MFH_ERR_INVALID, "null argument");
    require(c->dist.comm, MFH_ERR_STATE, "mfh_dist_setup has not run");
    auto &D = c->dist;
    mfh_comm *cm = D.comm;
    mfh_dist_stats st{};
    st.world = cm->world; st.rank = cm->rank; st.transport = D.transport; st.peer_enabled = cm->peer.enabled ? 1 : 0;
    st.halo_nodes_sent = D.sendPtr.back(); st.halo_nodes_received = D.recvPtr.back();
    st.halo_bytes_per_exchange = (D.sendPtr.back() + D.recvPtr.back()) * (int64_t)c->bs() * (int64_t)sizeof(double);
    st.interior_items = D.nInterior; st.boundary_items = D.nBoundary;
    st.exchange_ms = D.exchangeMs; st.interior_ms = D.interiorMs; st.boundary_ms = D.boundaryMs; st.exposed_wait_ms = D.exposedMs; st.operator_ms = D.operatorMs;
    st.profiled_applications = D.profiled;
    st.exchanges = D.nExchanges;
    st.peer_halo_messages = cm->peer.haloMessages; st.peer_halo_bytes = cm->peer.haloBytes;
    st.allreduces_small = cm->peer.smallAllreduces; st.allreduces_large = cm->peer.largeAllreduces;
    st.fallback_exchanges = cm->peer.fallbackExchanges; st.fallback_allreduces = cm->peer.fallbackAllreduces;
    *out = st;
    MFH_CATCH(c)
}

mfh_status mfh_comm_allreduce(mfh_ctx *c, mfh_comm *cm, double *dev, int64_t n) {
    MFH_TRY(c)
    require(c && cm && dev && n >= 0, MFH_ERR_INVALID, "bad arguments");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    comm_allreduce(cm, dev, n, c->stream);
    MFH_HIP(hipStreamSynchronize(c->stream));
    peer_check(cm, c->stream);
    MFH_CATCH(c)
}

// First contact with a transport must not be able to hang the run: the self test and the preflight wait for their stream with a deadline
// (MFH_COMM_TIMEOUT_S, default 120 s). When it passes, an RCCL communicator is aborted (ncclCommAbort ends its kernels, the stream drains) and
// the call fails like any other failed self test: the caller moves on to the next transport (distributed.py robust_comm). The peer transfers
// carry their own 60 s limit inside the kernels (mfh_peer.hip).
static void sync_with_deadline(mfh_comm *cm, hipStream_t s, const char *what) {
    static const double limit = [] { const char *e = getenv("MFH_COMM_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 120.0; }();
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return;
        if (q != hipErrorNotReady) { MFH_HIP(q); }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) break;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    (void)hipGetLastError();
    if (cm && cm->nccl && rccl().commAbort) {
        (void)rccl().commAbort(cm->nccl);
        cm->nccl = nullptr;                      // (aborted communicators are not destroyed again; the handle is gone)
        cm->aborted = true;
        (void)hipStreamSynchronize(s);
    } else {
        // Callbacks / peer transfers cannot be aborted from here: throwing with the stream still busy would let the buffers of the caller
        // be released -- and its host staging vectors destroyed -- under copies that are still in flight. Both transports end on their own (the peer
        // kernels carry a 60 s limit, the callbacks' process group its own timeout): wait for the stream to drain, THEN report.
        fprintf(stderr, "[meshfem_hip] communicator self-test: no answer within %d s (%s); this transport cannot be aborted -- waiting for its own time limit\n", (int)limit, what);
        if (hipStreamSynchronize(s) != hipSuccess) (void)hipGetLastError();
    }
    throw Error(MFH_ERR_HIP, std::string("communicator self-test: no answer within ") + std::to_string((int)limit) + " s (" + what + "); the transport is unusable");
}

mfh_status mfh_comm_selftest(mfh_ctx *c, mfh_comm *cm) {
    MFH_TRY(c)
    require(c && cm, MFH_ERR_INVALID, "bad arguments");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    // point-to-point: a ring shift of a 1024-double message (to itself at world 1); all-reduce: every rank contributes rank + 1 (and a
    // longer vector). Several rounds with different contents: the peer transfers alternate between two staging buffers.
    const int n = 1024, nLong = 3000;
    DBuf<double> a, b, v;
    a.alloc(n); b.alloc(n); v.alloc(nLong);
    std::vector<double> h((size_t)n), got((size_t)n), hv((size_t)nLong);
    const int32_t to = (cm->rank + 1) % cm->world, from = (cm->rank + cm->world - 1) % cm->world;
    if (cm->peer.enabled) peer_reserve(cm, n, 1, (1u << to) | (1u << from), s);
    const int rounds = cm->peer.enabled ? 6 : 1;
    for (int round = 0; round < rounds; ++round) {
        for (int i = 0; i < n; ++i) h[i] = (cm->rank + 1) * 1000.0 + i + 7.0 * round;
        MFH_HIP(hipMemcpyAsync(a.p, h.data(), n * sizeof(double), hipMemcpyHostToDevice, s));
        b.zero(s);
        if (cm->world > 1) {
            // the exchange is grouped per peer: send to `to`, receive from `from` (the same peer at world 2)
            const double *sb[2] = {a.p, nullptr};
            double *rb[2] = {nullptr, b.p};
            int32_t peers[2] = {to, from};
            int64_t sc[2] = {n, 0}, rc[2] = {0, n};
            if (to == from) { rb[0] = b.p; rc[0] = n; comm_exchange(cm, 1, peers, sb, sc, rb, rc, s); }
            else comm_exchange(cm, 2, peers, sb, sc, rb, rc, s);
        } else if (cm->nccl) {
            RcclApi &r = rccl();
            rccl_check(r.groupStart(), "ncclGroupStart");
            rccl_check(r.send(a.p, n, kNcclFloat64, to, cm->nccl, s), "ncclSend");
            rccl_check(r.recv(b.p, n, kNcclFloat64, from, cm->nccl, s), "ncclRecv");
            rccl_check(r.groupEnd(), "ncclGroupEnd");
        } else
            MFH_HIP(hipMemcpyAsync(b.p, a.p, n * sizeof(double), hipMemcpyDeviceToDevice, s));
        MFH_HIP(hipMemcpyAsync(got.data(), b.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
        sync_with_deadline(cm, s, "ring shift");
        for (int i = 0; i < n; ++i)
            if (got[i] != (from + 1) * 1000.0 + i + 7.0 * round) throw Error(MFH_ERR_HIP, "communicator self-test: point-to-point message corrupted");
        double one[2] = {cm->rank + 1.0 + round, 1.0};
        MFH_HIP(hipMemcpyAsync(a.p, one, 2 * sizeof(double), hipMemcpyHostToDevice, s));
        comm_allreduce(cm, a.p, 2, s);
        MFH_HIP(hipMemcpyAsync(one, a.p, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        for (int i = 0; i < nLong; ++i) hv[i] = (cm->rank + 1.0) * (i % 17) + round;
        MFH_HIP(hipMemcpyAsync(v.p, hv.data(), nLong * sizeof(double), hipMemcpyHostToDevice, s));
        comm_allreduce(cm, v.p, nLong, s);
        MFH_HIP(hipMemcpyAsync(hv.data(), v.p, nLong * sizeof(double), hipMemcpyDeviceToHost, s));
        sync_with_deadline(cm, s, "all-reduce");
        const double tri = cm->world * (cm->world + 1) / 2.0;
        if (one[0] != tri + (double)round * cm->world || one[1] != (double)cm->world) throw Error(MFH_ERR_HIP, "communicator self-test: all-reduce gives a wrong sum");
        for (int i = 0; i < nLong; ++i)
            if (hv[i] != tri * (i % 17) + (double)round * cm->world) throw Error(MFH_ERR_HIP, "communicator self-test: all-reduce of a vector gives a wrong sum");
    }
    peer_check(cm, s);
    MFH_CATCH(c)
}

// First contact with a node's devices and links, before any solve. Collective. out (PREFLIGHT_DOUBLES(world) doubles):
//   [0] world  [1] rank  [2] device  [3] device memory free (bytes)  [4] total  [5] arena: held  [6] arena: live
//   [7] bytes of the timed messages  [8] ring bandwidth on the transport underneath, GB/s per direction (send to rank + 1 while receiving
//   from rank - 1)  [9] the same through the peer transfers (0: not enabled / message too long for the staging)  [10] all-reduce of ones
//   through the communicator (must equal world)  [11] peer transfers enabled
//   [16 + r]           hipDeviceCanAccessPeer(own device, device of rank r): 1 / 0, 2 = the same device
//   [16 + world + r]   the IPC slab of rank r is mapped here: 1 / 0 (the result of hipIpcOpenMemHandle at mfh_comm_enable_peer)
mfh_status mfh_comm_preflight(mfh_ctx *c, mfh_comm *cm, int64_t messageBytes, double *out, int64_t nOut) {
    MFH_TRY(c)
    require(c && cm && out, MFH_ERR_INVALID, "bad arguments");
    require_device(c);
    const int world = cm->world, me = cm->rank;
    require(nOut >= 16 + 2 * (int64_t)world, MFH_ERR_INVALID, "mfh_comm_preflight: output needs 16 + 2 world doubles");
    MFH_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    for (int64_t i = 0; i < nOut; ++i) out[i] = 0.0;
    out[0] = world; out[1] = me; out[2] = c->device;
    size_t fr = 0, tot = 0;
    MFH_HIP(hipMemGetInfo(&fr, &tot));
    int64_t ar[8];
    mfh::device_arena_stats(c->device, ar);
    out[3] = (double)fr; out[4] = (double)tot; out[5] = (double)ar[0]; out[6] = (double)ar[1];
    // devices of all ranks: an all-reduce in which every rank fills its own slot
    DBuf<double> slots;
    slots.alloc((size_t)world + 2);
    std::vector<double> hs((size_t)world + 2, 0.0);
    hs[(size_t)me] = (double)c->device + 1.0;
    hs[(size_t)world] = 1.0;
    MFH_HIP(hipMemcpyAsync(slots.p, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice, s));
    comm_allreduce(cm, slots.p, world + 1, s);
    MFH_HIP(hipMemcpyAsync(hs.data(), slots.p, hs.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    sync_with_deadline(cm, s, "preflight");
    out[10] = hs[(size_t)world];
    out[11] = cm->peer.enabled ? 1.0 : 0.0;
    for (int r = 0; r < world; ++r) {
        const int devR = (int)hs[(size_t)r] - 1;
        int can = 0;
        if (devR == c->device) can = 2;
        else if (devR >= 0 && hipDeviceCanAccessPeer(&can, c->device, devR) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        out[16 + r] = can;
        out[16 + world + r] = (cm->peer.enabled && r < PEER_MAX_WORLD && cm->peer.remote[r]) ? 1.0 : 0.0;
    }
    // ring bandwidth: every rank sends messageBytes to rank + 1 and receives as much from rank - 1, `reps` times back to back
    const int64_t n = std::max<int64_t>(messageBytes / 8, 1024);
    out[7] = (double)(n * 8);
    if (world > 1) {
        DBuf<double> a, b;
        a.alloc((size_t)n); b.alloc((size_t)n);
        a.zero(s); b.zero(s);
        const int32_t to = (me + 1) % world, from = (me + world - 1) % world;
        const double *sb[2] = {a.p, nullptr};
        double *rb[2] = {nullptr, b.p};
        int32_t peers[2] = {to, from};
        int64_t sc[2] = {n, 0}, rc[2] = {0, n};
        int np = 2;
        if (to == from) { rb[0] = b.p; rc[0] = n; np = 1; }
        auto timed = [&](bool peerPath) {
            const int reps = 5;
            auto once = [&]() {
                if (peerPath) peer_exchange(cm, np, peers, sb, sc, rb, rc, s);
                else base_exchange(cm, np, peers, sb, sc, rb, rc, s);
            };
            once();                                        // warm-up: first touch of the links / of the callbacks' staging
            sync_with_deadline(cm, s, "preflight");
            const double t0 = now_ms();
            for (int k = 0; k < reps; ++k) once();
            sync_with_deadline(cm, s, "preflight");
            return (double)(n * 8) * reps / ((now_ms() - t0) * 1e-3) / 1e9;
        };
        out[8] = timed(false);
        if (cm->peer.enabled) {
            peer_reserve(cm, n, 1, (1u << to) | (1u << from), s);
            if (peer_can_exchange(cm, np, peers, sc, rc)) out[9] = timed(true);
            peer_check(cm, s);
        }
    }
    MFH_CATCH(c)
}

mfh_status mfh_dist_setup(mfh_ctx *c, mfh_comm *cm, int32_t nPeers, const int32_t *peers, const int64_t *sendPtr, const int32_t *sendNodes,
                          const int64_t *recvPtr) {
    MFH_TRY(c)
    require(c && c->haveMesh && cm && nPeers >= 0 && (nPeers == 0 || (peers && sendPtr && recvPtr)), MFH_ERR_INVALID, "bad distributed setup arguments");
    require(c->dofForNode.empty() || c->nOwnedDoFSet >= 0, MFH_ERR_UNSUPPORTED,
            "a row-partitioned context with a DoF map needs mfh_dof_map_partitioned (which DoFs are this rank's rows)");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    auto &D = c->dist;
    // block rows of the vectors: nodes, or DoFs when a DoF map is installed (the lists are in the same numbering)
    const int64_t nOwned = c->nOwnedDoF(), nHalo = c->nDoF - nOwned;
    D.peers.assign(peers, peers + nPeers);
    D.sendPtr.assign(1, 0); D.recvPtr.assign(1, 0);
    for (int k = 0; k < nPeers; ++k) {
        require(peers[k] >= 0 && peers[k] < cm->world && peers[k] != cm->rank, MFH_ERR_INVALID, "bad peer rank");
        require(sendPtr[k + 1] >= sendPtr[k] && recvPtr[k + 1] >= recvPtr[k], MFH_ERR_INVALID, "exchange offsets must be non-decreasing");
        D.sendPtr.push_back(sendPtr[k + 1]); D.recvPtr.push_back(recvPtr[k + 1]);
    }
    require(nPeers == 0 || (sendPtr[0] == 0 && recvPtr[0] == 0), MFH_ERR_INVALID, "exchange offsets start at 0");
    require(D.recvPtr.back() == nHalo, MFH_ERR_INVALID, "the receive ranges must cover the halo nodes exactly (halo nodes grouped by owner, in the peers' order)");
    std::vector<int32_t> idx(sendNodes, sendNodes + D.sendPtr.back());
    for (int32_t v : idx) require(v >= 0 && v < nOwned, MFH_ERR_INVALID, "send list holds a node this rank does not own");
    D.sendNodesHost = idx;
    D.sendIdx.upload(idx.empty() ? std::vector<int32_t>{0} : idx, c->stream);
    if (!D.commStream) {
        MFH_HIP(hipStreamCreateWithFlags(&D.commStream, hipStreamNonBlocking));
        for (auto &e : D.ev) MFH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (D.comm != cm) {
        dist_detach(c);
        std::lock_guard<std::mutex> lock(cm->mu);
        cm->users.push_back(c);
    }
    D.comm = cm;
    D.listKind = 0;
    D.sendBufW = std::max(D.sendBufW, c->bs());
    D.nExchanges = 0;
    if (cm->peer.enabled) {          // collective: every rank of the communicator is in mfh_dist_setup
        int64_t maxPair = 0;
        uint32_t mask = 0;
        for (int k = 0; k < nPeers; ++k) {
            maxPair = std::max(maxPair, std::max(sendPtr[k + 1] - sendPtr[k], recvPtr[k + 1] - recvPtr[k]));
            mask |= 1u << peers[k];
        }
        peer_reserve(cm, maxPair, c->bs(), mask, c->stream);
    }
    c->mg.valid = false;             // a multigrid hierarchy holds exchange lists derived from the previous ones
    MFH_CATCH(c)
}

mfh_status mfh_dist_two_level(mfh_ctx *c, int32_t nAgg, const int32_t *aggOfNode, const double *relPos) {
    MFH_TRY(c)
    require(c && c->dist.comm, MFH_ERR_STATE, "mfh_dist_setup has not run");
    const int64_t m = (int64_t)nAgg * (c->dim() == 3 ? 6 : 3);
    DBuf<double> Ac;
    Ac.alloc((size_t)m * m);
    mfh_status st = mfh_tl_partitioned_begin(c, nAgg, aggOfNode, relPos, Ac.p);
    if (st != MFH_OK) throw Error(st, c->err);
    DistLink L(c);
    L.allreduce(Ac.p, m * m);
    MFH_HIP(hipStreamSynchronize(c->stream));
    st = mfh_tl_partitioned_finish(c, Ac.p);
    if (st != MFH_OK) throw Error(st, c->err);
    c->precond = MFH_PRECOND_TWO_LEVEL;
    MFH_CATCH(c)
}

mfh_status mfh_dist_solve(mfh_ctx *c, int32_t nrhs, const double *f, double *u, double rtol, int32_t maxit, mfh_solve_info *info) {
    MFH_TRY(c)
    require(c && c->haveMesh && f && u && nrhs > 0 && maxit > 0 && rtol > 0, MFH_ERR_INVALID, "bad solve arguments");
    require(c->dist.comm, MFH_ERR_STATE, "mfh_dist_setup has not run");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    // Every decision that changes the SEQUENCE of collectives is agreed on before the first one of the solve: a
    // local error in the preparation, the operator in use, the lift of non-zero fixed values and the preconditioner. One
    // all-reduce of six flags; a rank that cannot run the cluster operator takes every rank to the assembled SpMV.
    dist_agree(c);
    c->dist.profUsed = 0;
    // MFH_PRECOND_MULTIGRID on every rank: the V-cycle with partitioned nodal levels and replicated aggregate levels (collective setup;
    // one right-hand side at a time in the classic loop)
    bool useMG = false;
    if (c->dist.multigridAgreed) useMG = ensure_multigrid(c) && c->mg.distributed == dist_active(c);
    else if (c->mg.valid && c->mg.distributed) destroy_multigrid(c);
    const int64_t n = (int64_t)c->bs() * c->sym.nRows;
    std::vector<mfh_solve_info> infos((size_t)nrhs);
    // every rank must take the same path: batches are chosen from (dim, nrhs) only
    int k0 = 0;
    while (k0 < nrhs) {
        int nb = 1;
        if (c->batchRhs && !useMG && !c->deterministic)
            for (int cand : {6, 3, 2})
                if (cand <= nrhs - k0 && k::op_batch_supported(c->bs(), cand)) { nb = cand; break; }
        if (nb == 1 && (c->distPcgVariant == 0 || useMG || c->deterministic)) solve_classic_partitioned(c, f + (size_t)k0 * n, u + (size_t)k0 * n, rtol, maxit, infos.data() + k0);
        else solve_cg(c, nb, f + (size_t)k0 * n, u + (size_t)k0 * n, n, rtol, maxit, infos.data() + k0);
        k0 += nb;
    }
    bool all = true;
    for (int k2 = 0; k2 < nrhs; ++k2) { all &= infos[k2].converged != 0; if (info) info[k2] = infos[k2]; }
    dist_profile_collect(c);
    peer_check(c->dist.comm, c->stream);
    if (!all) throw Error(MFH_ERR_NOT_CONVERGED, "PCG did not reach the requested tolerance within maxit iterations");
    MFH_CATCH(c)
}

mfh_status mfh_dist_apply_K(mfh_ctx *c, const double *uOwned, double *KuOwned) {
    MFH_TRY(c)
    require(c && c->haveMesh && uOwned && KuOwned, MFH_ERR_INVALID, "null argument");
    require(c->dist.comm, MFH_ERR_STATE, "mfh_dist_setup has not run");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    dist_agree(c);
    const int d = c->bs();
    const int64_t nOwn = c->sym.nRows * d, nAll = c->sym.nCols * d;
    if (c->cgU.n < (size_t)nAll) c->cgU.alloc((size_t)nAll);
    if (c->cgW.n < (size_t)nOwn) c->cgW.alloc((size_t)nOwn);
    DistLink L(c);
    MFH_HIP(hipMemcpyAsync(c->cgU.p, uOwned, (size_t)nOwn * sizeof(double), hipMemcpyHostToDevice, c->stream));
    apply_op_nr(c, L, 1, c->cgU.p, c->cgW.p, false, nullptr, nullptr, 0, nullptr);
    MFH_HIP(hipMemcpyAsync(KuOwned, c->cgW.p, (size_t)nOwn * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    MFH_HIP(hipStreamSynchronize(c->stream));
    MFH_CATCH(c)
}

mfh_status mfh_dev_memcpy(mfh_ctx *c, void *dst, const void *src, int64_t bytes, int32_t kind, void *stream) {
    MFH_TRY(c)
    require(c && dst && src && bytes >= 0 && kind >= 0 && kind <= 2, MFH_ERR_INVALID, "bad copy arguments");
    require_device(c);
    MFH_HIP(hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const hipMemcpyKind kd = kind == 0 ? hipMemcpyHostToDevice : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
    MFH_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, kd, s));
    MFH_HIP(hipStreamSynchronize(s));
    MFH_CATCH(c)
}

}   // extern "C"
