// Direct device-to-device transfers between the ranks of one node (HIP IPC): the halo exchange and the small all-reduces of the
// row-partitioned PCG without a communication library in the loop. See PeerState (mfh_comm.hh) for the protocol. The reference is
// single-process (no communication backend); this file has no counterpart to cite beyond north_star's
// "halo DOFs and dot products over xGMI".
//
// Memory model. The slabs are fine-grained device memory (hipDeviceMallocFinegrained: coherent between devices, the allocation
// kind RCCL uses for its own buffers); data is written with plain stores, followed by __threadfence_system() in every writing
// thread, a workgroup count, and ONE system-scope release store of the message number by the last workgroup. The receiver reads
// the number with system-scope acquire loads in a one-wave kernel; the copy out of the staging area is a later kernel on the
// same stream (kernel boundaries acquire at system scope as well).
#include "mfh_device.hh"
#include "mfh_comm.hh"

namespace mfh {

namespace {

struct PeerCopyArgs {
    int n;                                   // segments
    const double *src[PEER_MAX_WORLD];
    double *dst[PEER_MAX_WORLD];
    int64_t start[PEER_MAX_WORLD + 1];       // prefix sums of the segment lengths (doubles)
    int nSig;
    uint64_t *sig[PEER_MAX_WORLD];           // written by the last workgroup once every segment is out
    uint64_t sigVal[PEER_MAX_WORLD];
    unsigned *counter;                       // workgroup counter (this rank's slab)
};

struct PeerWaitArgs {
    int n;
    const uint64_t *flag[PEER_MAX_WORLD];
    uint64_t val[PEER_MAX_WORLD];
    int32_t who[PEER_MAX_WORLD];
    uint64_t *err;
    uint64_t timeoutTicks;
};

__device__ __forceinline__ void sys_store_release(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ uint64_t sys_load_acquire(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }

// spin until *flag >= val; a wait that outlasts the limit raises the error word (bit `who`) and returns: the solve then ends with
// garbage that the host reports (peer_check), the device never hangs
__device__ __forceinline__ void spin_until(const uint64_t *flag, uint64_t val, uint64_t *err, int who, uint64_t timeoutTicks) {
    const uint64_t t0 = wall_clock64();
    while (sys_load_acquire(flag) < val) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeoutTicks) {
            atomicOr((unsigned long long *)err, 1ull << (who & 63));
            break;
        }
    }
}

__global__ void __launch_bounds__(256) k_peer_copy(PeerCopyArgs a) {
    const int64_t total = a.start[a.n];
    int k = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        while (i >= a.start[k + 1]) ++k;
        a.dst[k][i - a.start[k]] = a.src[k][i - a.start[k]];
    }
    if (a.nSig == 0) return;
    __threadfence_system();
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(a.counter, 1u);
        last = t == gridDim.x - 1;
        if (last) *a.counter = 0u;
    }
    __syncthreads();
    if (last && (int)threadIdx.x < a.nSig) {
        __threadfence_system();
        sys_store_release(a.sig[threadIdx.x], a.sigVal[threadIdx.x]);
    }
}

__global__ void __launch_bounds__(64) k_peer_wait(PeerWaitArgs w) {
    if ((int)threadIdx.x < w.n) spin_until(w.flag[threadIdx.x], w.val[threadIdx.x], w.err, w.who[threadIdx.x], w.timeoutTicks);
}

// dev[i] <- sum over the ranks IN RANK ORDER (every rank gets the same bits): the own contribution from dev, the others from the staging area
__global__ void __launch_bounds__(256) k_peer_sum(double *__restrict__ dev, int64_t n, int world, int me, const double *__restrict__ stage, int64_t arCap) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double acc = 0.0;
        for (int r = 0; r < world; ++r) acc += r == me ? dev[i] : stage[(int64_t)r * arCap + i];
        dev[i] = acc;
    }
}

struct PeerSmallArgs {
    double *dev;
    int64_t n;
    int world, me;
    double *remoteSlot[PEER_MAX_WORLD];      // rank r's staging slot for this rank (parity applied)
    uint64_t *remoteFlag[PEER_MAX_WORLD];    // rank r's all-reduce word for this rank
    const uint64_t *localFlag[PEER_MAX_WORLD];
    const double *localStage;                // this rank's staging area of the parity: [world][arCap]
    int64_t arCap;
    uint64_t seq;
    uint64_t *err;
    uint64_t timeoutTicks;
};

// the whole all-reduce of a few doubles (PCG dot products) in ONE single-workgroup kernel: write to every rank, signal, wait, sum
__global__ void __launch_bounds__(256) k_peer_allreduce_small(PeerSmallArgs a) {
    for (int r = 0; r < a.world; ++r) {
        if (r == a.me) continue;
        for (int64_t i = threadIdx.x; i < a.n; i += 256) a.remoteSlot[r][i] = a.dev[i];
    }
    __threadfence_system();
    __syncthreads();
    const int r = threadIdx.x;
    if (r < a.world && r != a.me) {
        sys_store_release(a.remoteFlag[r], a.seq);
        spin_until(a.localFlag[r], a.seq, a.err, r, a.timeoutTicks);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    for (int64_t i = threadIdx.x; i < a.n; i += 256) {
        double acc = 0.0;
        for (int q = 0; q < a.world; ++q) acc += q == a.me ? a.dev[i] : a.localStage[(int64_t)q * a.arCap + i];
        a.dev[i] = acc;
    }
}

// ---- slab layout
size_t ctl_bytes() { return 4096; }
uint64_t *ctl_word(void *slab, int src, int w) { return reinterpret_cast<uint64_t *>(slab) + (size_t)src * PEER_CTL_WORDS + w; }
uint64_t *err_word(void *slab) { return reinterpret_cast<uint64_t *>(slab) + (size_t)PEER_MAX_WORLD * PEER_CTL_WORDS; }
// workgroup counters of the copy kernels (local words of the control page, zeroed with it): one per launch site -- halo push, halo pull,
// all-reduce push -- so that collectives of different kinds never count on the same word. Two collectives of the SAME kind on
// one communicator must still be ordered by the caller: a communicator serves one host thread / one solve at a time (meshfem_hip.h).
unsigned *counter_word(void *slab, int site = 0) { return reinterpret_cast<unsigned *>(reinterpret_cast<uint64_t *>(slab) + (size_t)PEER_MAX_WORLD * PEER_CTL_WORDS + 1 + site); }
double *halo_slot(const PeerState &P, void *slab, int parity, int src, int world) {
    return reinterpret_cast<double *>(reinterpret_cast<char *>(slab) + ctl_bytes()) + ((size_t)parity * world + src) * (size_t)P.haloCap;
}
double *ar_area(const PeerState &P, void *slab, int parity, int world) {
    return reinterpret_cast<double *>(reinterpret_cast<char *>(slab) + ctl_bytes()) + (size_t)2 * world * (size_t)P.haloCap + (size_t)parity * world * (size_t)P.arCap;
}
size_t slab_size(int world, int64_t haloCap, int64_t arCap) { return ctl_bytes() + (size_t)2 * world * ((size_t)haloCap + (size_t)arCap) * sizeof(double); }

uint64_t timeout_ticks(const PeerState &P, int device) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) khz = 100000;   // gfx9: 100 MHz
    return (uint64_t)(P.timeoutS * 1e3 * (double)khz);
}

// every rank contributes m doubles; out[r * m + j] = rank r's j-th value (the transport underneath sums a zero-padded table)
void allgather(mfh_comm *cm, const double *mine, int m, std::vector<double> &out, hipStream_t s) {
    const int world = cm->world;
    out.assign((size_t)world * m, 0.0);
    for (int j = 0; j < m; ++j) out[(size_t)cm->rank * m + j] = mine[j];
    DBuf<double> d;
    d.upload(out, s);
    base_allreduce(cm, d.p, (int64_t)world * m, s);
    d.download(out.data(), out.size(), s);
}

void close_remote(mfh_comm *cm) {
    PeerState &P = cm->peer;
    for (int r = 0; r < cm->world && r < PEER_MAX_WORLD; ++r) {
        if (r != cm->rank && P.remote[r]) (void)hipIpcCloseMemHandle(P.remote[r]);
        P.remote[r] = nullptr;
    }
}

}   // namespace

void peer_release(mfh_comm *cm) {
    PeerState &P = cm->peer;
    if (!P.slab) { P.enabled = false; return; }
    (void)hipDeviceSynchronize();
    close_remote(cm);
    (void)hipFree(P.slab);
    P.slab = nullptr;
    P.slabBytes = 0;
    P.enabled = false;
}

// Collective (every mfh_dist_setup): record the peer sets of all ranks and whether the largest halo message of ANY rank fits the staging
// (the answer is then the same everywhere). maxPairNodes: the largest number of block rows this rank sends to or receives from one
// peer; W: doubles per block row the exchanges may carry. The slabs are never reallocated: hipIpcGetMemHandle has been seen to refuse
// (invalid argument) an allocation made after another rank's mapping was closed and the own slab freed -- one allocation per
// communicator, sized at mfh_comm_enable_peer (MFH_PEER_HALO_CAP / MFH_PEER_AR_CAP doubles per pair and direction).
void peer_reserve(mfh_comm *cm, int64_t maxPairNodes, int W, uint32_t myPeerMask, hipStream_t s) {
    PeerState &P = cm->peer;
    if (!P.enabled) return;
    const int world = cm->world, me = cm->rank;
    MFH_HIP(hipSetDevice(cm->device));
    MFH_HIP(hipStreamSynchronize(s));
    P.peerMask[me] |= myPeerMask;
    double mine[2] = {(double)(maxPairNodes * (int64_t)W), (double)P.peerMask[me]};
    std::vector<double> all;
    allgather(cm, mine, 2, all, s);
    int64_t need = 0;
    for (int r = 0; r < world; ++r) {
        need = std::max<int64_t>(need, (int64_t)all[(size_t)r * 2]);
        P.peerMask[r] = (uint32_t)all[(size_t)r * 2 + 1];
    }
    P.haloNeed = std::max(P.haloNeed, need);
    P.symmetric = true;
    for (int r = 0; r < world; ++r)
        for (int q = 0; q < world; ++q)
            if (((P.peerMask[r] >> q) & 1u) != ((P.peerMask[q] >> r) & 1u)) P.symmetric = false;
}

// Collective: allocate, export and map the slabs.
void peer_enable(mfh_comm *cm, int device, hipStream_t s) {
    PeerState &P = cm->peer;
    if (P.enabled) return;
    if (cm->world < 2) throw Error(MFH_ERR_UNSUPPORTED, "peer transfers need at least two ranks");
    if (cm->world > PEER_MAX_WORLD) throw Error(MFH_ERR_UNSUPPORTED, "peer transfers: more than 16 ranks (HIP IPC does not leave the node)");
    if (cm->device < 0) cm->device = device;
    const int world = cm->world, me = cm->rank;
    if (const char *e = getenv("MFH_PEER_TIMEOUT_S")) P.timeoutS = std::max(1.0, atof(e));
    // halo messages: 2^22 doubles = 32 MiB per (buffer, source rank) -- the larger direction of a z-slab interface of BASELINE configs[4] is
    // 3.8e5 nodes x 3 doubles = 1.1 M; all-reduces: 2^21 doubles -- the replicated aggregate levels of the multigrid hierarchy
    // (6 x 262 144 doubles) fit; 96 MiB x world per rank in all
    P.haloCap = (int64_t)1 << 22;
    P.arCap = (int64_t)1 << 21;
    if (const char *e = getenv("MFH_PEER_HALO_CAP")) P.haloCap = std::max<int64_t>(1024, atoll(e));
    if (const char *e = getenv("MFH_PEER_AR_CAP")) P.arCap = std::max<int64_t>(PEER_AR_SMALL, atoll(e));
    MFH_HIP(hipSetDevice(cm->device));
    // the capacities come from the environment of every rank: agree before allocating
    {
        double mine[2] = {(double)P.haloCap, (double)P.arCap};
        std::vector<double> all;
        allgather(cm, mine, 2, all, s);
        for (int r = 0; r < world; ++r)
            if ((int64_t)all[(size_t)r * 2] != P.haloCap || (int64_t)all[(size_t)r * 2 + 1] != P.arCap)
                throw Error(MFH_ERR_INVALID, "peer transfers: MFH_PEER_HALO_CAP / MFH_PEER_AR_CAP differ between the ranks");
    }
    P.slabBytes = slab_size(world, P.haloCap, P.arCap);
    bool fine = true;
    std::string failure;
    // The protocol needs cross-device coherence (remote plain stores + a flag polled by a kernel that is already running, data read in
    // the same kernel): only fine-grained memory gives it. No coarse-grained fallback: without the fine-grained slab this
    // rank reports failure and EVERY rank backs out to the next transport.
    if (hipExtMallocWithFlags(&P.slab, P.slabBytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        fine = false;
        P.slab = nullptr;
        failure = "fine-grained allocation of the staging slab failed";
    }
    hipIpcMemHandle_t h;
    memset(&h, 0, sizeof(h));
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    if (P.slab) {
        MFH_HIP(hipMemsetAsync(P.slab, 0, ctl_bytes(), s));
        MFH_HIP(hipStreamSynchronize(s));
        hipError_t e = hipIpcGetMemHandle(&h, P.slab);
        if (e != hipSuccess) { (void)hipGetLastError(); failure = std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e); }
    }
    // handles + a failure flag travel together: every rank learns whether EVERY rank can go on (no rank is left waiting in a collective)
    double enc[66];
    for (int j = 0; j < 64; ++j) enc[j] = (double)reinterpret_cast<const unsigned char *>(&h)[j];
    enc[64] = fine ? 1.0 : 0.0;
    enc[65] = failure.empty() ? 0.0 : 1.0;
    std::vector<double> all;
    allgather(cm, enc, 66, all, s);
    bool anyFailed = false;
    for (int r = 0; r < world; ++r) anyFailed |= all[(size_t)r * 66 + 65] != 0.0;
    if (!anyFailed) {
        for (int r = 0; r < world && failure.empty(); ++r) {
            if (r == me) { P.remote[r] = P.slab; continue; }
            hipIpcMemHandle_t hr;
            for (int j = 0; j < 64; ++j) reinterpret_cast<unsigned char *>(&hr)[j] = (unsigned char)all[(size_t)r * 66 + j];
            // fault injection for the transport-chain test (tests/test_gpu_peer_transport.py): rank MFH_PEER_FAULT_RANK opens a handle whose
            // bytes were overwritten -- the open fails there, the second agreement round makes EVERY rank back out
            if (const char *fe = getenv("MFH_PEER_FAULT_RANK"))
                if (atoi(fe) == me) memset(&hr, 0x5a, sizeof(hr));
            hipError_t e = hipIpcOpenMemHandle(&P.remote[r], hr, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                P.remote[r] = nullptr;
                failure = std::string("hipIpcOpenMemHandle (rank ") + std::to_string(r) + "): " + hipGetErrorString(e) +
                          " -- the ranks must be separate processes on one node, HSA_ENABLE_IPC_MODE_LEGACY=0";
            }
        }
        // second round: did every rank map every slab?
        double ok = failure.empty() ? 0.0 : 1.0;
        allgather(cm, &ok, 1, all, s);
        for (int r = 0; r < world; ++r) anyFailed |= all[(size_t)r] != 0.0;
    }
    if (anyFailed) {
        close_remote(cm);
        if (P.slab) { (void)hipFree(P.slab); P.slab = nullptr; }
        throw Error(MFH_ERR_HIP, "peer transfers unavailable: " + (failure.empty() ? std::string("another rank could not export or map a staging slab") : failure));
    }
    for (auto &q : P.haloSeq) q = 0;
    P.arSeq = 0;
    P.haloNeed = 0;
    P.symmetric = true;
    for (auto &m : P.peerMask) m = 0;
    P.enabled = true;
    cm->descFull = cm->desc + " + peer copies over HIP IPC (fine-grained staging, " +
                   std::to_string(P.slabBytes >> 20) + " MiB per rank)";
}

bool peer_can_exchange(const mfh_comm *cm, int nPeers, const int32_t *peers, const int64_t *sendCounts, const int64_t *recvCounts) {
    const PeerState &P = cm->peer;
    if (!P.enabled || !P.slab) return false;
    // the same answer on every rank by construction: haloNeed is the largest pair of ANY rank at the width agreed in peer_reserve, the
    // symmetry of the peer sets was checked on the table every rank holds. The per-pair checks below cannot fail when those hold.
    if (!P.symmetric || P.haloNeed > P.haloCap) return false;
    for (int k = 0; k < nPeers; ++k) {
        if (!((P.peerMask[cm->rank] >> peers[k]) & 1u)) return false;
        if (sendCounts[k] > P.haloCap || recvCounts[k] > P.haloCap) return false;
    }
    return true;
}

void peer_exchange(mfh_comm *cm, int nPeers, const int32_t *peers, const double *const *sendBufs, const int64_t *sendCounts,
                   double *const *recvBufs, const int64_t *recvCounts, hipStream_t s) {
    PeerState &P = cm->peer;
    const int world = cm->world, me = cm->rank;
    if (nPeers == 0) return;
    PeerCopyArgs push{}, pull{};
    PeerWaitArgs w{};
    int64_t outTotal = 0, inTotal = 0;
    push.n = pull.n = nPeers;
    push.nSig = nPeers; pull.nSig = 0;
    push.counter = counter_word(P.slab, 0);
    pull.counter = counter_word(P.slab, 1);
    w.n = nPeers; w.err = err_word(P.slab); w.timeoutTicks = timeout_ticks(P, cm->device);
    push.start[0] = pull.start[0] = 0;
    for (int k = 0; k < nPeers; ++k) {
        const int q = peers[k];
        const uint64_t seq = ++P.haloSeq[q];
        const int parity = (int)(seq & 1);
        push.src[k] = sendBufs[k];
        push.dst[k] = halo_slot(P, P.remote[q], parity, me, world);
        push.start[k + 1] = push.start[k] + sendCounts[k];
        push.sig[k] = ctl_word(P.remote[q], me, 0);
        push.sigVal[k] = seq;
        w.flag[k] = ctl_word(P.slab, q, 0);
        w.val[k] = seq;
        w.who[k] = q;
        pull.src[k] = halo_slot(P, P.slab, parity, q, world);
        pull.dst[k] = recvBufs[k];
        pull.start[k + 1] = pull.start[k] + recvCounts[k];
        outTotal += sendCounts[k]; inTotal += recvCounts[k];
    }
    hipLaunchKernelGGL(k_peer_copy, dim3(k::grid_for(outTotal, 512)), dim3(256), 0, s, push);
    hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, s, w);
    if (inTotal > 0) hipLaunchKernelGGL(k_peer_copy, dim3(k::grid_for(inTotal, 512)), dim3(256), 0, s, pull);
    MFH_HIP(hipGetLastError());
    P.haloMessages += nPeers;
    P.haloBytes += (outTotal + inTotal) * (int64_t)sizeof(double);
}

bool peer_can_allreduce(const mfh_comm *cm, int64_t n) { return cm->peer.enabled && cm->peer.slab && n <= cm->peer.arCap; }

void peer_allreduce(mfh_comm *cm, double *dev, int64_t n, hipStream_t s) {
    PeerState &P = cm->peer;
    const int world = cm->world, me = cm->rank;
    const uint64_t seq = ++P.arSeq;
    const int parity = (int)(seq & 1);
    if (n <= PEER_AR_SMALL) {
        PeerSmallArgs a{};
        a.dev = dev; a.n = n; a.world = world; a.me = me;
        for (int r = 0; r < world; ++r) {
            a.remoteSlot[r] = ar_area(P, P.remote[r], parity, world) + (size_t)me * (size_t)P.arCap;
            a.remoteFlag[r] = ctl_word(P.remote[r], me, 1);
            a.localFlag[r] = ctl_word(P.slab, r, 1);
        }
        a.localStage = ar_area(P, P.slab, parity, world);
        a.arCap = P.arCap; a.seq = seq; a.err = err_word(P.slab); a.timeoutTicks = timeout_ticks(P, cm->device);
        hipLaunchKernelGGL(k_peer_allreduce_small, dim3(1), dim3(256), 0, s, a);
        MFH_HIP(hipGetLastError());
        ++P.smallAllreduces;
        return;
    }
    PeerCopyArgs push{};
    PeerWaitArgs w{};
    push.counter = counter_word(P.slab, 2);
    w.err = err_word(P.slab); w.timeoutTicks = timeout_ticks(P, cm->device);
    push.start[0] = 0;
    int k = 0;
    for (int r = 0; r < world; ++r) {
        if (r == me) continue;
        push.src[k] = dev;
        push.dst[k] = ar_area(P, P.remote[r], parity, world) + (size_t)me * (size_t)P.arCap;
        push.start[k + 1] = push.start[k] + n;
        push.sig[k] = ctl_word(P.remote[r], me, 1);
        push.sigVal[k] = seq;
        w.flag[k] = ctl_word(P.slab, r, 1);
        w.val[k] = seq;
        w.who[k] = r;
        ++k;
    }
    push.n = push.nSig = w.n = k;
    hipLaunchKernelGGL(k_peer_copy, dim3(k::grid_for(n * k, 1024)), dim3(256), 0, s, push);
    hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, s, w);
    hipLaunchKernelGGL(k_peer_sum, dim3(k::grid_for(n, 1024)), dim3(256), 0, s, dev, n, world, me, (const double *)ar_area(P, P.slab, parity, world), P.arCap);
    MFH_HIP(hipGetLastError());
    ++P.largeAllreduces;
}

void peer_check(mfh_comm *cm, hipStream_t s) {
    PeerState &P = cm->peer;
    if (!P.enabled || !P.slab) return;
    uint64_t e = 0;
    MFH_HIP(hipMemcpyAsync(&e, err_word(P.slab), sizeof(e), hipMemcpyDeviceToHost, s));
    MFH_HIP(hipStreamSynchronize(s));
    if (!e) return;
    MFH_HIP(hipMemsetAsync(err_word(P.slab), 0, sizeof(e), s));
    std::string who;
    for (int r = 0; r < cm->world; ++r) if ((e >> r) & 1ull) who += (who.empty() ? "" : ", ") + std::to_string(r);
    throw Error(MFH_ERR_HIP, "peer transfers: rank " + std::to_string(cm->rank) + " waited more than " + std::to_string((int)P.timeoutS) +
                                 " s for rank(s) " + who + " (a rank left the collective sequence, or the devices cannot see each other's memory)");
}

}   // namespace mfh
