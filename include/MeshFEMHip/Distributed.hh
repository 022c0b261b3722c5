////////////////////////////////////////////////////////////////////////////////
// MeshFEMHip/Distributed.hh
////////////////////////////////////////////////////////////////////////////////
// Header-only C++ facade of the row-partitioned (one process per GPU) solve of the C ABI (include/meshfem_hip.h, section
// "multi-GPU solve"). The reference is single-process (TBB, Parallelism.hh:31-43): there is no class to mirror, so the
// names follow the serial Simulator's where a method has a counterpart (solve, applyStiffnessMatrix, setIsotropicMaterial,
// LinearElasticity.hh:479-487,801-823) and describe the partition otherwise.
//
//   Communicator::uniqueId()                      rank 0 creates the RCCL id and distributes the 128 bytes (MPI_Bcast, a file ...)
//   Communicator::rccl(ctx, id, rank, world)      the library's own RCCL communicator (dlopen, no link-time dependency)
//   Communicator::callbacks(...)                  or two caller-supplied collectives (e.g. MPI_Allreduce / MPI_Sendrecv)
//   PartitionedSimulator<N, Deg> sim(device);
//   sim.setLocalMesh(elemNodes, nodePos, nOwned)  owned nodes first, halo nodes after them grouped by owner rank
//   sim.setExchange(comm, peers, sendPtr, sendNodes, recvPtr)
//   sim.solve(fOwned)                             Chronopoulos-Gear PCG inside the library: packed halo buffers, exchange
//                                                 overlapped with the interior element blocks, one all-reduce per iteration
#ifndef MESHFEMHIP_DISTRIBUTED_HH
#define MESHFEMHIP_DISTRIBUTED_HH

#include "LinearElasticity.hh"

namespace MeshFEMHip {

// RAII owner of one mfh_comm
class Communicator {
public:
    static mfh_rccl_unique_id uniqueId() {
        mfh_rccl_unique_id id;
        if (mfh_rccl_get_unique_id(&id) != MFH_OK) throw std::runtime_error("meshfem_hip: RCCL not found (mfh_rccl_get_unique_id)");
        return id;
    }
    static Communicator rccl(mfh_ctx *ctx, const mfh_rccl_unique_id &id, int rank, int world) {
        Communicator c;
        check(ctx, mfh_comm_create_rccl(ctx, &id, rank, world, &c.m_comm));
        c.m_rank = rank; c.m_world = world;
        return c;
    }
    static Communicator callbacks(int rank, int world, void *user, mfh_allreduce_fn allreduceSum, mfh_exchange_fn exchange) {
        Communicator c;
        if (mfh_comm_create_callbacks(rank, world, user, allreduceSum, exchange, &c.m_comm) != MFH_OK)
            throw std::runtime_error("meshfem_hip: bad communicator callbacks");
        c.m_rank = rank; c.m_world = world;
        return c;
    }
    Communicator(Communicator &&o) noexcept : m_comm(o.m_comm), m_rank(o.m_rank), m_world(o.m_world) { o.m_comm = nullptr; }
    Communicator &operator=(Communicator &&o) noexcept {
        if (this != &o) { mfh_comm_destroy(m_comm); m_comm = o.m_comm; m_rank = o.m_rank; m_world = o.m_world; o.m_comm = nullptr; }
        return *this;
    }
    Communicator(const Communicator &) = delete;
    Communicator &operator=(const Communicator &) = delete;
    ~Communicator() { mfh_comm_destroy(m_comm); }
    mfh_comm *get() const { return m_comm; }
    int rank() const { return m_rank; }
    int world() const { return m_world; }
    std::string describe() const { return mfh_comm_describe(m_comm); }
    // ring shift + all-reduce with known answers over the actual transport
    void selfTest(mfh_ctx *ctx) const { check(ctx, mfh_comm_selftest(ctx, m_comm)); }
    // Direct peer transfers on top of this communicator (collective; the ranks are separate processes of one node): halos and small
    // all-reduces become device stores into the neighbours' staging slabs mapped through HIP IPC. Returns false -- on EVERY rank -- when
    // they cannot be set up (one rank, no IPC on this host): the communicator underneath stays in use, why() says why.
    bool enablePeerTransfers(mfh_ctx *ctx, std::string *why = nullptr) {
        if (mfh_comm_enable_peer(ctx, m_comm) == MFH_OK) return true;
        if (why) *why = mfh_last_error(ctx);
        return false;
    }
    void disablePeerTransfers(mfh_ctx *ctx) { check(ctx, mfh_comm_disable_peer(ctx, m_comm)); }
private:
    Communicator() = default;
    mfh_comm *m_comm = nullptr;
    int m_rank = 0, m_world = 1;
};

namespace LinearElasticity {

// One rank's share of a row-partitioned elasticity problem. Vectors are flattened [x0 y0 z0 x1 ...] over the OWNED nodes.
template <size_t N, size_t Deg>
class PartitionedSimulator {
public:
    explicit PartitionedSimulator(int device = 0) : m_owner(device) {}
    mfh_ctx *ctx() const { return m_owner.get(); }

    // elemNodes: nodesPerElem local node ids per element (MeshFEM local order), every element with at least one owned node;
    // nodePos: N coordinates per local node; the first nOwned nodes are this rank's rows
    void setLocalMesh(const std::vector<int32_t> &elemNodes, const std::vector<Real> &nodePos, size_t nOwned) {
        const size_t npe = N == 3 ? (Deg == 1 ? 4 : 10) : (Deg == 1 ? 3 : 6);
        if (elemNodes.size() % npe || nodePos.size() % N) throw std::runtime_error("setLocalMesh: bad array sizes");
        m_numLocal = nodePos.size() / N;
        m_numOwned = nOwned;
        check(ctx(), mfh_mesh_set(ctx(), (int32_t)N, (int32_t)Deg, (int64_t)(elemNodes.size() / npe), (int64_t)m_numLocal, (int64_t)nOwned,
                                  elemNodes.data(), nodePos.data()));
    }
    void setIsotropicMaterial(Real E, Real nu) { check(ctx(), mfh_material_isotropic(ctx(), E, nu)); }
    void setMaterial(const std::vector<Real> &D) { check(ctx(), mfh_material_const(ctx(), D.data())); }
    // fixed variables N * localNode + c, for halo nodes as well as owned ones (SPSDSystem::fixVariables semantics)
    void fixVariables(const std::vector<size_t> &vars, const std::vector<Real> &vals = std::vector<Real>()) {
        std::vector<int64_t> v(vars.begin(), vars.end());
        if (!vals.empty() && vals.size() != vars.size()) throw std::runtime_error("Incorrect number of fixedVarValues");
        check(ctx(), mfh_fix_variables(ctx(), (int64_t)v.size(), v.data(), vals.empty() ? nullptr : vals.data()));
    }
    // peers[k] reads our owned nodes sendNodes[sendPtr[k] .. sendPtr[k+1]) and owns our halo nodes nOwned + [recvPtr[k], recvPtr[k+1])
    void setExchange(const Communicator &comm, const std::vector<int32_t> &peers, const std::vector<int64_t> &sendPtr,
                     const std::vector<int32_t> &sendNodes, const std::vector<int64_t> &recvPtr) {
        if (sendPtr.size() != peers.size() + 1 || recvPtr.size() != peers.size() + 1) throw std::runtime_error("setExchange: bad offsets");
        check(ctx(), mfh_dist_setup(ctx(), comm.get(), (int32_t)peers.size(), peers.data(), sendPtr.data(), sendNodes.data(), recvPtr.data()));
    }
    // two-level preconditioner on GLOBAL aggregates: aggregate id and (position - aggregate centre) / H per local node
    void setTwoLevelPreconditioner(int nAggregates, const std::vector<int32_t> &aggOfNode, const std::vector<Real> &relPos) {
        if (aggOfNode.size() != m_numLocal || relPos.size() != 3 * m_numLocal) throw std::runtime_error("setTwoLevelPreconditioner: one entry per local node");
        check(ctx(), mfh_dist_two_level(ctx(), nAggregates, aggOfNode.data(), relPos.data()));
    }
    // multigrid V-cycle (call it on EVERY rank): nodal levels partitioned like the mesh, aggregate levels replicated; the
    // hierarchy is built collectively inside the next solve. Needs no aggregates from the caller.
    void setMultigridPreconditioner() { check(ctx(), mfh_set_preconditioner(ctx(), MFH_PRECOND_MULTIGRID)); }
    std::vector<Real> solve(const std::vector<Real> &fOwned) {
        if (fOwned.size() != N * m_numOwned) throw std::runtime_error("solve: the load has N entries per owned node");
        std::vector<Real> u(fOwned.size());
        check(ctx(), mfh_dist_solve(ctx(), 1, fOwned.data(), u.data(), rtol, maxit, &info));
        return u;
    }
    std::vector<Real> applyStiffnessMatrix(const std::vector<Real> &uOwned) {
        std::vector<Real> Ku(uOwned.size());
        check(ctx(), mfh_dist_apply_K(ctx(), uOwned.data(), Ku.data()));
        return Ku;
    }
    // transport in use, halo sizes, exchange / overlap timings (option "dist_profile"), message counters: mfh_dist_stats
    mfh_dist_stats stats() { mfh_dist_stats st{}; check(ctx(), mfh_dist_get_stats(ctx(), &st)); return st; }
    // run-to-run bit-reproducible operator and dot products (all-reduces in rank order need the peer transfers or a reproducible transport)
    void setDeterministic(bool on) { check(ctx(), mfh_set_option(ctx(), "deterministic", on ? 1.0 : 0.0)); }
    size_t numOwnedNodes() const { return m_numOwned; }
    size_t numLocalNodes() const { return m_numLocal; }
    double rtol = 1e-8;
    int maxit = 100000;
    mfh_solve_info info{};
private:
    Context m_owner;
    size_t m_numOwned = 0, m_numLocal = 0;
};

}   // namespace LinearElasticity
}   // namespace MeshFEMHip
#endif
