// INTEGRATION.md section 1b, compiled: the call sequence a MeshFEM maintainer puts into the reference's own Simulator to keep its mesh /
// boundary-condition code and replace only m_assembleStiffnessMatrix + the factorisation (LinearElasticity.hh:1377-1404,1408-1466).
// `HostSim` stands for that Simulator: it owns a node table in ITS numbering (here: the FEM nodes of a small cantilever, permuted), per-element
// flattened tensors, a DoF map and its Dirichlet variables, and hands exactly those to the C ABI:
//   mfh_mesh_set, mfh_material_tensor_field, mfh_dof_map, mfh_assemble, mfh_clear_fixed, mfh_fix_variables, mfh_solve, mfh_export_upper_triplets.
// argv[1] = device ordinal (-1: host-only context -> the solve must fail loudly)
#include <meshfem_hip.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

static void ck(mfh_ctx *c, mfh_status st) {
    if (st != MFH_OK) throw std::runtime_error(c ? mfh_last_error(c) : "no context");
}

struct HostSim {                      // what the reference's Simulator already holds
    std::vector<int32_t> elemNodes;   // element(i).node(j).index(), 10 per quadratic tet
    std::vector<double> nodePos;      // node(i)->p
    std::vector<double> flattenedD;   // per element 6 x 6 (ETensorStoreGetter -> ElasticityTensor::D(i, j))
    std::vector<int32_t> dofForNode;  // m_dofForNode (identity here)
    std::vector<int64_t> fixedVars;   // m_getDirichletVarsAndValues
    std::vector<double> fixedVals;
    size_t numNodes() const { return nodePos.size() / 3; }
    size_t numElements() const { return elemNodes.size() / 10; }
};

int main(int argc, char **argv) {
    const int device = argc > 1 ? atoi(argv[1]) : 0;
    try {
        // ---- a mesh in "the reference's" numbering: build the FEM node table once with the library on a host-only context, then permute it
        const int nx = 3;
        std::vector<double> V;
        for (int k = 0; k <= 1; ++k) for (int j = 0; j <= 1; ++j) for (int i = 0; i <= nx; ++i) { V.push_back(i); V.push_back(j); V.push_back(k); }
        auto id = [&](int i, int j, int k) { return (int32_t)(i + (nx + 1) * (j + 2 * k)); };
        std::vector<int32_t> T;
        const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
        for (int i = 0; i < nx; ++i)
            for (auto &p : perm) {
                int c[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {1, 1, 1}};
                c[1][p[0]] = 1; c[2][p[0]] = 1; c[2][p[1]] = 1;
                int32_t t[4];
                for (int q = 0; q < 4; ++q) t[q] = id(i + c[q][0], c[q][1], c[q][2]);
                const double *a = &V[3 * t[0]], *b = &V[3 * t[1]], *cc = &V[3 * t[2]], *dd = &V[3 * t[3]];
                double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, v[3] = {cc[0] - a[0], cc[1] - a[1], cc[2] - a[2]}, w[3] = {dd[0] - a[0], dd[1] - a[1], dd[2] - a[2]};
                double det = u[0] * (v[1] * w[2] - v[2] * w[1]) - u[1] * (v[0] * w[2] - v[2] * w[0]) + u[2] * (v[0] * w[1] - v[1] * w[0]);
                if (det < 0) std::swap(t[0], t[1]);
                T.insert(T.end(), t, t + 4);
            }
        HostSim sim;
        {
            mfh_ctx *h = nullptr;
            ck(nullptr, mfh_create(-1, &h));
            ck(h, mfh_mesh_build(h, 3, 2, (int64_t)T.size() / 4, (int64_t)V.size() / 3, T.data(), V.data()));
            int64_t nE = 0, nN = 0;
            ck(h, mfh_mesh_sizes(h, &nE, &nN, nullptr, nullptr, nullptr, nullptr, nullptr));
            std::vector<int32_t> en((size_t)nE * 10);
            std::vector<double> pos((size_t)nN * 3);
            ck(h, mfh_mesh_get_elem_nodes(h, en.data()));
            ck(h, mfh_mesh_get_node_positions(h, pos.data()));
            mfh_destroy(h);
            std::vector<int32_t> newId((size_t)nN);                 // the host code's own numbering: reversed
            for (int64_t n = 0; n < nN; ++n) newId[(size_t)n] = (int32_t)(nN - 1 - n);
            sim.nodePos.resize(pos.size());
            for (int64_t n = 0; n < nN; ++n) for (int c = 0; c < 3; ++c) sim.nodePos[(size_t)newId[(size_t)n] * 3 + c] = pos[(size_t)n * 3 + c];
            sim.elemNodes.resize(en.size());
            for (size_t k = 0; k < en.size(); ++k) sim.elemNodes[k] = newId[(size_t)en[k]];
        }
        const double E = 200.0, nu = 0.35, lam = nu * E / ((1 + nu) * (1 - 2 * nu)), mu = E / (2 + 2 * nu);
        sim.flattenedD.assign(sim.numElements() * 36, 0.0);
        for (size_t e = 0; e < sim.numElements(); ++e) {
            double *D = &sim.flattenedD[e * 36];
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) D[i * 6 + j] = lam + (i == j ? 2 * mu : 0.0);
            for (int i = 3; i < 6; ++i) D[i * 6 + i] = mu;
        }
        sim.dofForNode.resize(sim.numNodes());
        std::iota(sim.dofForNode.begin(), sim.dofForNode.end(), 0);
        std::vector<double> f(3 * sim.numNodes(), 0.0);
        for (size_t n = 0; n < sim.numNodes(); ++n) {
            if (std::fabs(sim.nodePos[3 * n]) < 1e-12) for (int c = 0; c < 3; ++c) { sim.fixedVars.push_back((int64_t)(3 * n + c)); sim.fixedVals.push_back(0.0); }
            if (std::fabs(sim.nodePos[3 * n] - nx) < 1e-12) f[3 * n + 1] = -0.01;
        }
        printf("host simulator: %zu elements, %zu nodes, %zu fixed variables\n", sim.numElements(), sim.numNodes(), sim.fixedVars.size());

        // ---- m_buildConstrainedSystem, replaced
        mfh_ctx *hip = nullptr;
        if (mfh_create(device, &hip) != MFH_OK) throw std::runtime_error("no usable HIP device (there is no CPU fallback)");
        ck(hip, mfh_mesh_set(hip, 3, 2, (int64_t)sim.numElements(), (int64_t)sim.numNodes(), (int64_t)sim.numNodes(), sim.elemNodes.data(), sim.nodePos.data()));
        ck(hip, mfh_material_tensor_field(hip, sim.flattenedD.data()));
        ck(hip, mfh_dof_map(hip, sim.dofForNode.data(), (int64_t)sim.numNodes()));
        ck(hip, mfh_assemble(hip, MFH_ASSEMBLE_GATHER));
        ck(hip, mfh_clear_fixed(hip));
        ck(hip, mfh_fix_variables(hip, (int64_t)sim.fixedVars.size(), sim.fixedVars.data(), sim.fixedVals.data()));
        // ---- solve(f), replaced
        std::vector<double> u(f.size());
        mfh_solve_info info{};
        ck(hip, mfh_solve(hip, 1, f.data(), u.data(), 1e-10, 100000, &info));
        // ---- dumpSystem / the matrix the reference would have handed to CHOLMOD
        uint64_t nnz = 0;
        ck(hip, mfh_export_upper_triplets(hip, nullptr, nullptr, nullptr, &nnz));
        std::vector<uint64_t> ti(nnz), tj(nnz);
        std::vector<double> tv(nnz);
        ck(hip, mfh_export_upper_triplets(hip, ti.data(), tj.data(), tv.data(), &nnz));
        // residual of the returned displacement on the free variables, with the exported matrix
        std::vector<double> Ku(f.size(), 0.0);
        for (uint64_t k = 0; k < nnz; ++k) {
            Ku[ti[k]] += tv[k] * u[tj[k]];
            if (ti[k] != tj[k]) Ku[tj[k]] += tv[k] * u[ti[k]];
        }
        std::vector<char> fixed(f.size(), 0);
        for (int64_t v : sim.fixedVars) fixed[(size_t)v] = 1;
        double rr = 0, ff = 0, tip = 0;
        for (size_t q = 0; q < f.size(); ++q) {
            if (!fixed[q]) { rr += (Ku[q] - f[q]) * (Ku[q] - f[q]); ff += f[q] * f[q]; }
            if (q % 3 == 1) tip = std::min(tip, u[q]);
        }
        printf("PCG %d iterations, |K u - f| / |f| on the free variables with the exported triplets = %.2e, tip deflection %.5f, K nnz %llu\n", info.iterations,
               std::sqrt(rr / ff), tip, (unsigned long long)nnz);
        mfh_destroy(hip);
        if (!(std::sqrt(rr / ff) < 1e-8) || !(tip < -1e-3)) return 2;
        printf("solver-only swap ok\n");
        return 0;
    } catch (const std::runtime_error &e) {
        printf("runtime_error: %s\n", e.what());
        return 3;
    }
}
