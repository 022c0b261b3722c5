// C++ drop-in test of the facade beyond the cantilever: PoissonMesh / Laplacian / mass matrix on the scalar
// path, the rigid-motion constraint (no_rigid_motion) on a free body, SPSDSystem on a caller-supplied matrix.
// argv[1] = device ordinal.
#include <MeshFEMHip/LinearElasticity.hh>
#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace MeshFEMHip;

static void kuhn_bar(int nx, std::vector<std::array<Real, 3>> &V, std::vector<std::array<int32_t, 4>> &T) {
    for (int k = 0; k <= 1; ++k) for (int j = 0; j <= 1; ++j) for (int i = 0; i <= nx; ++i) V.push_back({(Real)i, (Real)j, (Real)k});
    auto id = [&](int i, int j, int k) { return (int32_t)(i + (nx + 1) * (j + 2 * k)); };
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
}

#define REQUIRE(cond)                                                   \
    do {                                                                \
        if (!(cond)) { printf("FAILED: %s (line %d)\n", #cond, __LINE__); return 2; } \
    } while (0)

int main(int argc, char **argv) {
    const int device = argc > 1 ? atoi(argv[1]) : 0;
    const int nx = 4;
    std::vector<std::array<Real, 3>> V;
    std::vector<std::array<int32_t, 4>> T;
    kuhn_bar(nx, V, T);
    try {
        // ---- Poisson: u = 0 on x = 0, u = 8 on x = 4  =>  u = 2 x, grad u = (2, 0, 0)
        PoissonMesh<3, 2> pm(T, V, device);
        pm.applyDirichletBox({-1e-9, -9, -9}, {1e-9, 9, 9}, 0.0);
        pm.applyDirichletBox({nx - 1e-9, -9, -9}, {nx + 1e-9, 9, 9}, 2.0 * nx);
        std::vector<Real> u;
        pm.solve(u);
        auto g = pm.gradUAverage(u);
        Real gerr = 0;
        for (auto &x : g) gerr = std::max(gerr, std::max(std::fabs(x[0] - 2.0), std::max(std::fabs(x[1]), std::fabs(x[2]))));
        REQUIRE(u.size() == pm.numNodes() && gerr < 1e-8);
        // ---- mass matrix: 1^T M 1 = volume; Laplacian: rows sum to zero
        TripletMatrix Mm = pm.massMatrix(), L = pm.laplacian();
        Real msum = 0;
        for (auto &t : Mm.nz) msum += (t.i == t.j ? 1.0 : 2.0) * t.v;
        std::vector<Real> rowsum(L.m, 0.0);
        for (auto &t : L.nz) { rowsum[t.i] += t.v; if (t.i != t.j) rowsum[t.j] += t.v; }
        Real rmax = 0;
        for (Real r : rowsum) rmax = std::max(rmax, std::fabs(r));
        REQUIRE(std::fabs(msum - (Real)nx) < 1e-12 && rmax < 1e-12);
        printf("poisson: nodes %zu grad err %.2e iterations %d | mass sum %.12f | laplacian row sums %.1e\n", u.size(), gerr,
               pm.info.iterations, msum, rmax);

        // ---- free body with the rigid-motion constraint: pull on both ends, no Dirichlet condition at all
        LinearElasticity::Simulator<3, 2> sim(T, V, device);
        sim.setIsotropicMaterial(200.0, 0.35);
        sim.applyNeumannBox({nx - 1e-9, -9, -9}, {nx + 1e-9, 9, 9}, {1, 0, 0});
        sim.applyNeumannBox({-1e-9, -9, -9}, {1e-9, 9, 9}, {-1, 0, 0});
        bool threw = false;
        try { sim.solve(); } catch (const std::runtime_error &e) { threw = std::string(e.what()) == "Unimplemented"; }
        REQUIRE(threw);                                   // assembleConstrainedSystem's behaviour without constraints (:1240)
        sim.applyNoRigidMotionConstraint();
        sim.rtol = 1e-11;
        auto w = sim.solve();
        Real sum[3] = {0, 0, 0}, stretch = 0;
        for (size_t n = 0; n < w.size(); ++n) for (int a = 0; a < 3; ++a) sum[a] += w[n][a];
        for (size_t n = 0; n < V.size(); ++n) if (V[n][0] == nx) stretch = std::max(stretch, w[n][0]);
        // uniaxial tension: elongation = sigma L / E = 4 / 200, split symmetrically by the zero-mean constraint
        REQUIRE(std::fabs(sum[0]) < 1e-9 && std::fabs(sum[1]) < 1e-9 && std::fabs(sum[2]) < 1e-9);
        REQUIRE(std::fabs(stretch - 0.5 * nx / 200.0) < 1e-8);
        printf("free body: sum u = (%.1e %.1e %.1e), end displacement %.8f, iterations %d\n", sum[0], sum[1], sum[2], stretch,
               sim.info.iterations);

        // ---- SPSDSystem on a caller-supplied matrix: 1D Laplacian, both ends fixed -> linear profile
        const size_t n = 50;
        TripletMatrix K;
        K.m = K.n = n;
        for (size_t i = 0; i < n; ++i) {
            K.nz.push_back({i, i, (i == 0 || i + 1 == n) ? 1.0 : 2.0});
            if (i + 1 < n) K.nz.push_back({i, i + 1, -1.0});
        }
        GenericSPSDSystem sys(K, device);
        sys.fixVariables({0, n - 1}, {1.0, 50.0});
        std::vector<Real> b(n, 0.0), x;
        sys.solve(b, x);
        Real lerr = 0;
        for (size_t i = 0; i < n; ++i) lerr = std::max(lerr, std::fabs(x[i] - (1.0 + i)));
        REQUIRE(lerr < 1e-7);
        printf("generic SPSDSystem: max error %.2e iterations %d\n", lerr, sys.info.iterations);
        return 0;
    } catch (const std::runtime_error &e) {
        printf("runtime_error: %s\n", e.what());
        return 3;
    }
}
