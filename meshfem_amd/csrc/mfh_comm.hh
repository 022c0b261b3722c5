// The communicator behind the opaque mfh_comm handle (mfh_solver.cpp: RCCL / caller callbacks; mfh_peer.hip: direct
// device-to-device transfers over HIP IPC layered on top of either). Not part of the ABI.
#pragma once
#include "mfh_internal.hh"
#include <mutex>

struct mfh_ctx;

namespace mfh {

constexpr int PEER_MAX_WORLD = 16;      // ranks of one node (HIP IPC does not leave it)
constexpr int PEER_CTL_WORDS = 4;       // control words per source rank: halo READY, all-reduce READY, 2 spare
constexpr int64_t PEER_AR_SMALL = 512;  // all-reduces of at most this many doubles run as ONE single-workgroup kernel

// Direct transfers between the ranks' devices (option of an mfh_comm, mfh_comm_enable_peer). Every rank owns ONE slab of
// fine-grained device memory, exported with hipIpcGetMemHandle and mapped by every other rank:
//   ctl   uint64 [world][PEER_CTL_WORDS]  written remotely: ctl[src][0] = number of halo messages src has delivered here,
//                                         ctl[src][1] = number of all-reduce contributions; then local words (error flag, block counter)
//   halo  double [2][world][haloCap]      staging of the halo messages, by parity of the pair's message number and source rank
//   ar    double [2][world][arCap]        staging of the all-reduce contributions
// A message is written by the SENDER's kernel straight into the receiver's slab (stores over xGMI), followed by a
// system-scope release store of the message number; the receiver's stream waits for that number in a one-wave kernel and copies
// the staged entries to where they belong. Two buffers per pair make acknowledgements unnecessary: a rank starts message s only
// after it has finished s - 1, for which it has seen the peer's s - 1, which the peer sent after unpacking s - 2 (one stream, in
// order) -- the buffer of parity s is free. No host code, no library call in the loop.
struct PeerState {
    bool enabled = false;
    void *slab = nullptr;                        // this rank's slab
    size_t slabBytes = 0;
    int64_t haloCap = 0, arCap = 0;              // doubles per (parity, source rank)
    int64_t haloNeed = 0;                        // largest halo message (doubles) any rank has registered (peer_reserve)
    bool symmetric = true;                       // the registered peer sets are symmetric (q is a peer of r exactly when r is one of q)
    void *remote[PEER_MAX_WORLD] = {nullptr};    // the other ranks' slabs as mapped here (remote[rank] = slab)
    uint64_t haloSeq[PEER_MAX_WORLD] = {0};      // messages exchanged with every rank so far
    uint64_t arSeq = 0;                          // all-reduces so far
    uint32_t peerMask[PEER_MAX_WORLD] = {0};     // peer sets of all ranks as agreed at the last bootstrap (bit q of peerMask[r])
    double timeoutS = 60.0;                      // a wait that lasts longer raises the slab's error word instead of hanging the device
    std::string why;                             // why peer transfers are not in use (enable failed / refused)
    // statistics of the last solve
    int64_t haloMessages = 0, haloBytes = 0, smallAllreduces = 0, largeAllreduces = 0, fallbackExchanges = 0, fallbackAllreduces = 0;
};

}   // namespace mfh

struct mfh_comm {
    int rank = 0, world = 1;
    // callbacks
    void *user = nullptr;
    mfh_allreduce_fn allreduce = nullptr;
    mfh_exchange_fn exchange = nullptr;
    // RCCL
    void *nccl = nullptr;
    bool aborted = false;           // the self test gave up on the RCCL communicator (ncclCommAbort): every further use fails
    int device = -1;
    std::string desc, descFull;
    std::vector<mfh_ctx *> users;   // contexts whose mfh_dist_setup named this communicator (detached when it is destroyed)
    std::mutex mu;
    mfh::PeerState peer;
};

namespace mfh {
// the transport underneath (RCCL or the caller's callbacks); mfh_solver.cpp
void base_allreduce(mfh_comm *cm, double *dev, int64_t n, hipStream_t s);
void base_exchange(mfh_comm *cm, int nPeers, const int32_t *peers, const double *const *sendBufs, const int64_t *sendCounts,
                   double *const *recvBufs, const int64_t *recvCounts, hipStream_t s);
// mfh_peer.hip
void peer_enable(mfh_comm *cm, int device, hipStream_t s);                                   // collective
void peer_reserve(mfh_comm *cm, int64_t maxPairNodes, int W, uint32_t myPeerMask, hipStream_t s);   // collective (every mfh_dist_setup)
bool peer_can_exchange(const mfh_comm *cm, int nPeers, const int32_t *peers, const int64_t *sendCounts, const int64_t *recvCounts);
void peer_exchange(mfh_comm *cm, int nPeers, const int32_t *peers, const double *const *sendBufs, const int64_t *sendCounts,
                   double *const *recvBufs, const int64_t *recvCounts, hipStream_t s);
bool peer_can_allreduce(const mfh_comm *cm, int64_t n);
void peer_allreduce(mfh_comm *cm, double *dev, int64_t n, hipStream_t s);
void peer_check(mfh_comm *cm, hipStream_t s);      // throws if a wait timed out (blocking read of the error word)
void peer_release(mfh_comm *cm);
}   // namespace mfh
