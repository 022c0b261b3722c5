// The C++ side of the multi-GPU path without any Python or torch in the process: the library's RCCL communicator (found
// with dlopen in the ROCm installation), created at world size 1 from a locally generated unique id, its self test (grouped
// ncclSend / ncclRecv to itself + ncclAllReduce), and LinearElasticity::PartitionedSimulator::solve against the serial
// Simulator::solve of the same cantilever.
// argv[1] = device ordinal (-1: host-only context -> must throw, proving there is no fallback)
#include <MeshFEMHip/Distributed.hh>
#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace MeshFEMHip;

int main(int argc, char **argv) {
    const int device = argc > 1 ? atoi(argv[1]) : 0;
    const int nx = 6;
    std::vector<std::array<Real, 3>> V;
    for (int k = 0; k <= 1; ++k) for (int j = 0; j <= 1; ++j) for (int i = 0; i <= nx; ++i) V.push_back({(Real)i, (Real)j, (Real)k});
    auto id = [&](int i, int j, int k) { return (int32_t)(i + (nx + 1) * (j + 2 * k)); };
    std::vector<std::array<int32_t, 4>> T;
    const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    for (int i = 0; i < nx; ++i)
        for (auto &p : perm) {
            int c[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {1, 1, 1}};
            c[1][p[0]] = 1; c[2][p[0]] = 1; c[2][p[1]] = 1;
            std::array<int32_t, 4> t;
            for (int q = 0; q < 4; ++q) t[q] = id(i + c[q][0], c[q][1], c[q][2]);
            auto &a = V[t[0]], &b = V[t[1]], &cc = V[t[2]], &d = V[t[3]];
            Real u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, v[3] = {cc[0] - a[0], cc[1] - a[1], cc[2] - a[2]},
                 w[3] = {d[0] - a[0], d[1] - a[1], d[2] - a[2]};
            Real det = u[0] * (v[1] * w[2] - v[2] * w[1]) - u[1] * (v[0] * w[2] - v[2] * w[0]) + u[2] * (v[0] * w[1] - v[1] * w[0]);
            if (det < 0) std::swap(t[0], t[1]);
            T.push_back(t);
        }
    try {
        // serial reference: FEM node numbering, load, Dirichlet variables and solution from the ordinary Simulator
        LinearElasticity::Simulator<3, 2> ser(T, V, device);
        ser.setIsotropicMaterial(200.0, 0.35);
        ser.applyDirichletBox({-1e-4, -1e-4, -1e-4}, {1e-4, 1.0001, 1.0001}, {0, 0, 0}, true);
        ser.applyNeumannBox({0.9999, -1e-4, -1e-4}, {1.0001, 1.0001, 1.0001}, {0, -1, 0}, MFH_NEUMANN_FORCE, true);
        auto f = ser.neumannLoad();
        ser.rtol = 1e-10;
        auto uRef = ser.solve();
        const size_t nNode = ser.numNodes();
        std::vector<int32_t> en(T.size() * 10);
        std::vector<Real> pos(nNode * 3);
        check(ser.ctx(), mfh_mesh_get_elem_nodes(ser.ctx(), en.data()));
        check(ser.ctx(), mfh_mesh_get_node_positions(ser.ctx(), pos.data()));
        int64_t nFixed = 0;
        check(ser.ctx(), mfh_bc_dirichlet_vars(ser.ctx(), nullptr, nullptr, &nFixed));
        std::vector<int64_t> fv((size_t)nFixed);
        std::vector<Real> fx((size_t)nFixed);
        check(ser.ctx(), mfh_bc_dirichlet_vars(ser.ctx(), fv.data(), fx.data(), &nFixed));

        // "partition" over one rank: every node owned, no peers; the transport is still the library's RCCL communicator
        LinearElasticity::PartitionedSimulator<3, 2> sim(device);
        sim.setLocalMesh(en, pos, nNode);
        sim.setIsotropicMaterial(200.0, 0.35);
        const mfh_rccl_unique_id uid = Communicator::uniqueId();
        Communicator comm = Communicator::rccl(sim.ctx(), uid, 0, 1);
        printf("communicator: %s\n", comm.describe().c_str());
        comm.selfTest(sim.ctx());
        std::string why;
        if (comm.enablePeerTransfers(sim.ctx(), &why)) return 5;        // one rank: refused with a reason, the communicator stays as it is
        printf("peer transfers with one rank: %s\n", why.c_str());
        if (why.find("two ranks") == std::string::npos) return 6;
        sim.setExchange(comm, {}, {0}, {}, {0});
        sim.fixVariables(std::vector<size_t>(fv.begin(), fv.end()), fx);
        std::vector<Real> fFlat(3 * nNode);
        for (size_t n = 0; n < nNode; ++n) for (int c = 0; c < 3; ++c) fFlat[3 * n + c] = f[n][c];
        sim.rtol = 1e-10;
        auto u = sim.solve(fFlat);
        Real err = 0, nrm = 0;
        for (size_t n = 0; n < nNode; ++n)
            for (int c = 0; c < 3; ++c) { err += (u[3 * n + c] - uRef[n][c]) * (u[3 * n + c] - uRef[n][c]); nrm += uRef[n][c] * uRef[n][c]; }
        printf("partitioned solve: %d iterations, true residual %.2e, rel-L2 difference to the serial solve %.2e\n", sim.info.iterations,
               sim.info.true_rel_residual, std::sqrt(err / nrm));
        if (!sim.info.converged || std::sqrt(err / nrm) > 1e-7) return 2;
        // K u = f on the free variables
        auto Ku = sim.applyStiffnessMatrix(u);
        std::vector<char> fixed(3 * nNode, 0);
        for (auto v2 : fv) fixed[(size_t)v2] = 1;
        Real r = 0, fn = 0;
        for (size_t i = 0; i < Ku.size(); ++i) if (!fixed[i]) { r += (Ku[i] - fFlat[i]) * (Ku[i] - fFlat[i]); fn += fFlat[i] * fFlat[i]; }
        if (std::sqrt(r / fn) > 1e-8) return 4;
        const mfh_dist_stats st = sim.stats();
        printf("stats: world %d, transport %d, peer %d, exchanges %lld\n", st.world, st.transport, st.peer_enabled, (long long)st.exchanges);
        if (st.world != 1 || st.peer_enabled != 0) return 7;
        // bit-reproducible mode through the facade: two solves, identical bits
        sim.setDeterministic(true);
        auto d1 = sim.solve(fFlat), d2 = sim.solve(fFlat);
        if (d1 != d2) return 8;
        sim.setDeterministic(false);
        deviceCacheTrim();
        printf("distributed facade ok\n");
        return 0;
    } catch (const std::runtime_error &e) {
        printf("runtime_error: %s\n", e.what());
        return 3;
    }
}
