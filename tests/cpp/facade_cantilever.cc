// C++ drop-in smoke test of the facade: the cantilever of examples/cantilever (one hex column),
// written the way a Simulate_cli-style driver uses LinearElasticity::Simulator.
// argv[1] = device ordinal (-1: host-only context -> the solve must throw, proving there is no fallback)
#include <MeshFEMHip/LinearElasticity.hh>
#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace MeshFEMHip;

int main(int argc, char **argv) {
    const int device = argc > 1 ? atoi(argv[1]) : 0;
    // 4 x 1 x 1 hexes -> 24 tets each (the same split as hex_tet_subdiv), built by hand: 5 tets per
    // hex is enough here: use the 6-tet Kuhn split to keep the file short
    const int nx = 4;
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
            // orient positively
            auto &a = V[t[0]], &b = V[t[1]], &cc = V[t[2]], &d = V[t[3]];
            Real u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, v[3] = {cc[0] - a[0], cc[1] - a[1], cc[2] - a[2]},
                 w[3] = {d[0] - a[0], d[1] - a[1], d[2] - a[2]};
            Real det = u[0] * (v[1] * w[2] - v[2] * w[1]) - u[1] * (v[0] * w[2] - v[2] * w[0]) + u[2] * (v[0] * w[1] - v[1] * w[0]);
            if (det < 0) std::swap(t[0], t[1]);
            T.push_back(t);
        }
    try {
        LinearElasticity::Simulator<3, 2> sim(T, V, device);
        sim.setIsotropicMaterial(200.0, 0.35);
        sim.applyDirichletBox({-1e-4, -1e-4, -1e-4}, {1e-4, 1.0001, 1.0001}, {0, 0, 0}, true);
        sim.applyNeumannBox({0.9999, -1e-4, -1e-4}, {1.0001, 1.0001, 1.0001}, {0, -1, 0}, MFH_NEUMANN_FORCE, true);
        auto f = sim.neumannLoad();
        Real fy = 0;
        for (auto &x : f) fy += x[1];
        auto u = sim.solve();
        Real tip = 0;
        for (size_t n = 0; n < u.size(); ++n) tip = std::min(tip, u[n][1]);
        TripletMatrix K;
        sim.m_assembleStiffnessMatrix(K);
        // Euler-Bernoulli: delta = F L^3 / (3 E I), I = 1/12 -> 64 / (3*200/12) = 1.28 (+ shear)
        printf("nodes %zu total load %.6f tip deflection %.6f PCG iterations %d K nnz %zu\n", u.size(), fy, tip,
               sim.info.iterations, K.nnz());
        if (std::fabs(fy + 1.0) > 1e-12 || tip > -1.2 || tip < -1.6) return 2;
        auto eps = sim.averageStrainField(u);
        auto Ku = sim.applyStiffnessMatrix(u);
        // per-element strain interpolants: their corner mean is the averaged strain
        auto sf = sim.strainField(u);
        Real dev = 0;
        for (size_t e2 = 0; e2 < eps.size(); ++e2)
            for (int c2 = 0; c2 < 6; ++c2) {
                Real m = 0;
                for (int k2 = 0; k2 < 4; ++k2) m += sf[(e2 * 4 + k2) * 6 + c2] / 4;
                dev = std::max(dev, std::fabs(m - eps[e2][c2]));
            }
        if (dev > 1e-12) return 5;
        // uniform dilation by 2 through updateMeshNodePositions. The stored boundary TRACTION (force / old area) is kept, as
        // in the reference, so the total force grows by 4 while delta ~ F L^3 / (E I) ~ F / length: the deflection doubles
        auto V2 = V;
        for (auto &x : V2) for (auto &c2 : x) c2 *= 2.0;
        sim.updateMeshNodePositions(V2);
        auto u2 = sim.solve();
        Real tip2 = 0;
        for (size_t n = 0; n < u2.size(); ++n) tip2 = std::min(tip2, u2[n][1]);
        printf("after dilation by 2: tip deflection %.6f (expected %.6f)\n", tip2, 2 * tip);
        if (std::fabs(tip2 - 2 * tip) > 1e-6 * std::fabs(tip)) return 6;
        // shape derivative under the uniform dilation delta_p = x: K scales with length^(dim-2), so (delta K) u = K u in 3D
        decltype(u) dp(V.size());
        for (size_t n = 0; n < V.size(); ++n) dp[n] = {V[n][0], V[n][1], V[n][2]};
        auto dKu = sim.applyDeltaStiffnessMatrix(u, dp);
        Real err = 0, nrm = 0;
        for (size_t n = 0; n < Ku.size(); ++n)
            for (int c2 = 0; c2 < 3; ++c2) { err = std::max(err, std::fabs(dKu[n][c2] - Ku[n][c2])); nrm = std::max(nrm, std::fabs(Ku[n][c2])); }
        printf("dilation shape derivative: |dK u - K u| / |K u| = %.2e\n", err / nrm);
        if (err > 1e-10 * nrm) return 4;
        return 0;
    } catch (const std::runtime_error &e) {
        printf("runtime_error: %s\n", e.what());
        return 3;
    }
}
