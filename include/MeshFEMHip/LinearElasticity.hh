////////////////////////////////////////////////////////////////////////////////
// MeshFEMHip/LinearElasticity.hh
////////////////////////////////////////////////////////////////////////////////
// Header-only C++ facade over the C ABI (include/meshfem_hip.h) that keeps the reference's class
// and method names for the hot path, so that a Simulate_cli-style driver only swaps the include
// and the namespace:
//
//     MeshFEM                                            this facade
//     LinearElasticity::Simulator<Mesh<N,Deg,...>>       MeshFEMHip::LinearElasticity::Simulator<N,Deg>
//       Simulator(elems, vertices)        (:460-473)       same (throws std::runtime_error on inverted elements)
//       solve(f) / solve()                (:479-487,:657)  same, returns a per-node VField
//       neumannLoad(), constantStrainLoad (:703-717,:551)  same
//       applyPeriodicConditions(eps)      (:845-854)       same
//       applyBoundaryConditions(conds)    (:881-1027)      same, conds from readBoundaryConditions (BoundaryConditions.hh)
//       applyTranslationPins, applyPeriodicPairDirichletConditions, reportRegionSurfaceForces (:1087-1111,:1251-1270)  same
//       setMaterial(Materials::Constant)  (Simulate_cli.cc:104-175)   same (Materials.hh reads the .material JSON)
//       m_assembleStiffnessMatrix(K)      (:1408-1466)     same (upper triplets, summed)
//       applyStiffnessMatrix(u)           (:801-823)       same
//       averageStrainField/StressField    (:528-549)       same
//       applyDeltaStiffnessMatrix, deltaConstantStrainLoad, deltaAverageStrainField (:1301-1374)   same
//     SPSDSystem<Real>                                   MeshFEMHip::SPSDSystem
//       fixVariables(vars, vals), solve(f, u)  (SparseMatrices.hh:2389-2606)   same
//
// (line numbers: /root/reference/src/lib/MeshFEM/LinearElasticity.hh). Errors of the C ABI are
// rethrown as std::runtime_error, the reference's error convention. Eigen is not required: fields
// are std::vector<double> in the reference's interleaved layout (Fields.hh:15-17); with Eigen
// available `Eigen::Map<Eigen::VectorXd>(v.data(), v.size())` adapts them in place.
#ifndef MESHFEMHIP_LINEARELASTICITY_HH
#define MESHFEMHIP_LINEARELASTICITY_HH

#include <array>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../meshfem_hip.h"
#include "BoundaryConditions.hh"
#include "Materials.hh"

namespace MeshFEMHip {

using Real = double;   // Types.hh:8

inline void check(mfh_ctx *ctx, mfh_status st) {
    if (st != MFH_OK) throw std::runtime_error(ctx ? mfh_last_error(ctx) : "meshfem_hip: invalid context");
}

// RAII owner of one mfh_ctx (one per host thread / device)
class Context {
public:
    explicit Context(int device = 0) {
        if (mfh_create(device, &m_ctx) != MFH_OK)
            throw std::runtime_error("meshfem_hip: no usable HIP device (there is no CPU fallback)");
    }
    ~Context() { mfh_destroy(m_ctx); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    mfh_ctx *get() const { return m_ctx; }
private:
    mfh_ctx *m_ctx = nullptr;
};

// The per-process cache of released device blocks (meshfem_hip.h, "Device memory"): hand everything back to the driver, e.g. before
// another library or process needs the device's memory
inline void deviceCacheTrim() { mfh_device_cache_trim(); }

// Triplet / TripletMatrix with the reference's field names (SparseMatrices.hh:45-72,191-773)
struct Triplet { size_t i, j; Real v; };
struct TripletMatrix {
    size_t m = 0, n = 0;
    std::vector<Triplet> nz;
    size_t nnz() const { return nz.size(); }
    // u64 nnz, u64 i[], u64 j[], f64 v[]   (SparseMatrices.hh:629-645)
    void dumpBinary(const std::string &path) const {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("Couldn't open output file " + path);
        uint64_t cnt = nz.size();
        fwrite(&cnt, sizeof(cnt), 1, f);
        for (int pass = 0; pass < 2; ++pass)
            for (const auto &t : nz) { uint64_t x = pass == 0 ? t.i : t.j; fwrite(&x, sizeof(x), 1, f); }
        for (const auto &t : nz) fwrite(&t.v, sizeof(Real), 1, f);
        fclose(f);
    }
};

// SPSDSystem over an assembled context (SPD branch; SparseMatrices.hh:2321-2716)
class SPSDSystem {
public:
    explicit SPSDSystem(mfh_ctx *ctx) : m_ctx(ctx) { check(m_ctx, mfh_clear_fixed(m_ctx)); }
    void fixVariables(const std::vector<size_t> &fixedVars, const std::vector<Real> &fixedVarValues = std::vector<Real>()) {
        if (fixedVars.empty()) return;
        if (!fixedVarValues.empty() && fixedVarValues.size() != fixedVars.size())
            throw std::runtime_error("Incorrect number of fixedVarValues");
        std::vector<int64_t> v(fixedVars.begin(), fixedVars.end());
        check(m_ctx, mfh_fix_variables(m_ctx, (int64_t)v.size(), v.data(), fixedVarValues.empty() ? nullptr : fixedVarValues.data()));
    }
    template <class Vec, class SolnVec> void solve(const Vec &f, SolnVec &u) {
        std::vector<Real> fv(f.begin(), f.end()), uv(fv.size());
        check(m_ctx, mfh_solve(m_ctx, 1, fv.data(), uv.data(), rtol, maxit, &info));
        u.assign(uv.begin(), uv.end());
    }
    double rtol = 1e-8;   // PCG relative residual (replaces CHOLMOD's direct solve)
    int maxit = 100000;
    mfh_solve_info info{};
private:
    mfh_ctx *m_ctx;
};

namespace LinearElasticity {

template <size_t N, size_t Deg>
class Simulator {
public:
    using VField = std::vector<std::array<Real, N>>;     // per node / per DoF vectors (Fields.hh VectorField)
    using SMField = std::vector<std::array<Real, N *(N + 1) / 2>>;

    // elems: K+1 vertex ids per element; vertices: N coordinates each   (LinearElasticity.hh:460-473)
    Simulator(const std::vector<std::array<int32_t, N + 1>> &elems, const std::vector<std::array<Real, N>> &vertices, int device = 0)
        : m_owner(device) {
        check(ctx(), mfh_mesh_build(ctx(), (int32_t)N, (int32_t)Deg, (int64_t)elems.size(), (int64_t)vertices.size(),
                                    &elems[0][0], &vertices[0][0]));
        int64_t nn = 0;
        check(ctx(), mfh_mesh_sizes(ctx(), &m_numElements, &nn, nullptr, nullptr, nullptr, nullptr, nullptr));
        m_numNodes = m_numDoFs = (size_t)nn;
    }

    mfh_ctx *ctx() const { return m_owner.get(); }
    size_t numNodes() const { return m_numNodes; }
    size_t numDoFs() const { return m_numDoFs; }
    size_t numElements() const { return (size_t)m_numElements; }

    // materials: Materials::Constant / per-element fields (Simulate_cli.cc:104-175)
    void setIsotropicMaterial(Real E, Real nu) { check(ctx(), mfh_material_isotropic(ctx(), E, nu)); }
    void setMaterialTensor(const std::vector<Real> &D /* flatLen^2 row-major */) { check(ctx(), mfh_material_const(ctx(), D.data())); }
    void setIsotropicField(const std::vector<Real> &E, const std::vector<Real> &nu) {
        if (E.size() != numElements() || nu.size() != numElements()) throw std::runtime_error("setIsotropicField: one (E, nu) pair per element expected");
        check(ctx(), mfh_material_iso_field(ctx(), E.data(), nu.data()));
    }
    void setOrthotropicField(const std::vector<Real> &params) {
        if (params.size() != numElements() * (N == 3 ? 9 : 4)) throw std::runtime_error("setOrthotropicField: 9 (3D) / 4 (2D) parameters per element expected");
        check(ctx(), mfh_material_ortho_field(ctx(), params.data()));
    }

    void setMaterial(const Materials::Constant<N> &mat) { setMaterialTensor(mat.getTensor().flat()); }

    // ---- boundary conditions as the reference's condition objects (applyBoundaryConditions, :881-1027). Regions are
    // matched here on the host against the mesh tables of the context (boundary nodes / boundary elements by vertex
    // barycentre), expression values are evaluated per matched node / element, and the outcome is handed to the C ABI as
    // node lists (mfh_bc_dirichlet_nodes: "Condition applied to non-boundary node", "Conflicting dirichlet displacements."),
    // boundary-element tractions (mfh_bc_neumann_elements) and nodal forces (mfh_bc_delta_force).
    void applyBoundaryConditions(const std::vector<CondPtr<N>> &conds) {
        if (conds.empty()) return;
        const Geometry &g = m_geometry();
        ExpressionEnvironment env;
        BBox<N> mbb = boundingBox();
        env.setVectorValue("mesh_size_", mbb.dimensions());
        env.setVectorValue("mesh_min_", mbb.minCorner);
        env.setVectorValue("mesh_max_", mbb.maxCorner);
        const size_t nBE = g.area.size(), npbe = nBE ? g.beNodes.size() / nBE : 0;
        auto beCentre = [&](size_t b) {
            VectorND<N> c{};
            for (size_t k = 0; k < N; ++k) for (size_t d = 0; d < N; ++d) c[d] += g.pos[(size_t)g.beNodes[b * npbe + k]][d];
            for (size_t d = 0; d < N; ++d) c[d] /= (Real)N;
            return c;
        };
        auto setDirichlet = [&](const std::vector<int64_t> &nodes, const std::vector<VectorND<N>> &vals, const ComponentMask &mask) {
            if (nodes.empty()) return;
            check(ctx(), mfh_bc_dirichlet_nodes(ctx(), (int64_t)nodes.size(), nodes.data(), &vals[0][0], mask.bits(N)));
        };
        auto setTractions = [&](const std::vector<int64_t> &bes, const std::vector<VectorND<N>> &t) {
            if (bes.empty()) return;
            check(ctx(), mfh_bc_neumann_elements(ctx(), (int64_t)bes.size(), bes.data(), &t[0][0]));
        };
        for (const auto &cond : conds) {
            env.setVectorValue("region_size_", cond->region->dimensions());
            env.setVectorValue("region_min_", cond->region->minCorner);
            env.setVectorValue("region_max_", cond->region->maxCorner);
            if (auto nc = dynamic_cast<const NeumannCondition<N> *>(cond.get())) {
                Real regionArea = 0;
                std::vector<int64_t> region;
                std::vector<VectorND<N>> tractions;
                for (size_t b = 0; b < nBE; ++b) {
                    VectorND<N> centre = beCentre(b);
                    if (!nc->containsPoint(centre)) continue;
                    env.setXYZ(centre);
                    regionArea += g.area[b];
                    region.push_back((int64_t)b);
                    VectorND<N> t;
                    if (nc->type == NeumannType::Pressure) {
                        Real p = nc->pressure(env);
                        for (size_t d = 0; d < N; ++d) t[d] = -p * g.normal[b * N + d];
                    } else t = nc->traction(env);
                    tractions.push_back(t);
                }
                if (region.empty()) throw std::runtime_error("Neumann region unmatched");
                if (nc->type == NeumannType::Force)                 // a total force, spread uniformly over the region
                    for (auto &t : tractions) for (auto &x : t) x /= regionArea;
                setTractions(region, tractions);
            } else if (dynamic_cast<const TargetCondition<N> *>(cond.get()) || dynamic_cast<const TargetNodesCondition<N> *>(cond.get())) {
                fprintf(stderr, "WARNING: ignoring target boundary conditions.\n");
            } else if (auto dc = dynamic_cast<const DirichletCondition<N> *>(cond.get())) {
                ++m_dirichletRegionCount;
                std::vector<int64_t> nodes;
                std::vector<VectorND<N>> vals;
                for (int32_t n : g.bdryNodes) {
                    const VectorND<N> &p = g.pos[(size_t)n];
                    if (!dc->containsPoint(p)) continue;
                    env.setXYZ(p);
                    nodes.push_back(n);
                    vals.push_back(dc->displacement(env));
                    m_dirichletRegionOfNode[(size_t)n] = m_dirichletRegionCount;
                }
                setDirichlet(nodes, vals, dc->componentMask);
            } else if (auto dec = dynamic_cast<const DirichletElementsCondition<N> *>(cond.get())) {
                ++m_dirichletRegionCount;
                std::vector<int64_t> nodes;
                std::vector<VectorND<N>> vals;
                for (size_t b = 0; b < nBE; ++b) {
                    IVectorND<N> idx;
                    for (size_t k = 0; k < N; ++k) idx[k] = (size_t)g.beNodes[b * npbe + k];
                    if (!dec->containsElement(idx)) continue;
                    for (size_t k = 0; k < npbe; ++k) {
                        int32_t n = g.beNodes[b * npbe + k];
                        env.setXYZ(g.pos[(size_t)n]);
                        nodes.push_back(n);
                        vals.push_back(dec->displacement(env));
                        m_dirichletRegionOfNode[(size_t)n] = m_dirichletRegionCount;
                    }
                }
                setDirichlet(nodes, vals, dec->componentMask);
            } else if (auto nec = dynamic_cast<const NeumannElementsCondition<N> *>(cond.get())) {
                size_t numSet = 0;
                Real regionArea = 0;
                std::vector<int64_t> bes;
                std::vector<VectorND<N>> tractions;
                std::vector<size_t> forceEntries;
                for (size_t b = 0; b < nBE; ++b) {
                    UnorderedTriplet elem((size_t)g.beNodes[b * npbe], (size_t)g.beNodes[b * npbe + 1], N == 3 ? (size_t)g.beNodes[b * npbe + 2] : 0);
                    if (!nec->hasValueForElement(elem)) continue;
                    const auto &val = nec->getValue(elem);
                    VectorND<N> t;
                    if (val.type == NeumannType::Pressure) for (size_t d = 0; d < N; ++d) t[d] = -val.pressure() * g.normal[b * N + d];
                    else t = val.traction();
                    if (val.type == NeumannType::Force) { regionArea += g.area[b]; forceEntries.push_back(bes.size()); }
                    bes.push_back((int64_t)b);
                    tractions.push_back(t);
                    ++numSet;
                }
                if (numSet != nec->numElements()) throw std::runtime_error("Some element boundary conditions weren't matched.");
                for (size_t e : forceEntries) for (auto &x : tractions[e]) x /= regionArea;
                setTractions(bes, tractions);
            } else if (auto dnc = dynamic_cast<const DirichletNodesCondition<N> *>(cond.get())) {
                fprintf(stderr, "WARNING: dirichlet region index currently not set for DirichletNodesCondition; region force printout will be inaccurate.\n");
                std::vector<int64_t> nodes(dnc->indices.begin(), dnc->indices.end());
                setDirichlet(nodes, dnc->displacements, dnc->componentMask);
            } else if (auto fc = dynamic_cast<const DeltaForceCondition<N> *>(cond.get())) {
                for (size_t n = 0; n < g.pos.size(); ++n) {
                    if (!fc->containsPoint(g.pos[n])) continue;
                    env.setXYZ(g.pos[n]);
                    VectorND<N> f = fc->force(env);
                    check(ctx(), mfh_bc_delta_force(ctx(), (int64_t)n, f.data()));
                }
            } else if (auto fnc = dynamic_cast<const DeltaForceNodesCondition<N> *>(cond.get())) {
                for (size_t i = 0; i < fnc->indices.size(); ++i) {
                    size_t ni = fnc->indices[i];
                    if (ni > m_numNodes) throw std::runtime_error("DeltaForceNodesCondition node index out of bounds: " + std::to_string(ni));
                    check(ctx(), mfh_bc_delta_force(ctx(), (int64_t)ni, fnc->forces[i].data()));
                }
            } else throw std::runtime_error("Illegal BC type");
        }
    }
    // component d of the boundary node with the smallest coordinate d is pinned to zero (:1095-1111)
    void applyTranslationPins(const ComponentMask &c) {
        const Geometry &g = m_geometry();
        for (size_t d = 0; d < N; ++d) {
            if (!c.has(d) || g.bdryNodes.empty()) continue;
            int32_t best = g.bdryNodes[0];
            for (int32_t n : g.bdryNodes) if (g.pos[(size_t)n][d] < g.pos[(size_t)best][d]) best = n;
            ComponentMask dmask;
            dmask.set(d);
            int64_t node = best;
            VectorND<N> zero{};
            check(ctx(), mfh_bc_dirichlet_nodes(ctx(), 1, &node, zero.data(), dmask.bits(N)));
        }
    }
    void applyPeriodicPairDirichletConditions(std::vector<PeriodicPairDirichletCondition<N>> &pps) {     // :1087-1093
        const Geometry &g = m_geometry();
        std::vector<VectorND<N>> bpos;
        for (int32_t n : g.bdryNodes) bpos.push_back(g.pos[(size_t)n]);
        BBox<N> bb = boundingBox();
        for (auto &pp : pps) {
            auto p = pp.pair(bpos, bb);
            int64_t nodes[2] = {g.bdryNodes[p.first], g.bdryNodes[p.second]};
            std::array<VectorND<N>, 2> zero{};
            check(ctx(), mfh_bc_dirichlet_nodes(ctx(), 2, nodes, &zero[0][0], pp.component().bits(N)));
        }
    }
    void removeBoundaryConditions() {                            // mfh_bc_clear: Dirichlet, Neumann and delta forces
        check(ctx(), mfh_bc_clear(ctx()));
        m_dirichletRegionOfNode.clear();
        m_dirichletRegionCount = 0;
    }
    // K u summed over the boundary nodes of every Dirichlet region; entry 0 collects the boundary nodes outside all of
    // them (reportRegionSurfaceForces, :1251-1270)
    std::vector<VectorND<N>> regionSurfaceForces(const VField &uNodes) const {
        const Geometry &g = m_geometry();
        if (m_numDoFs != m_numNodes) throw std::runtime_error("regionSurfaceForces: not available under periodic conditions");
        VField f = applyStiffnessMatrix(uNodes);
        std::vector<VectorND<N>> forces(m_dirichletRegionCount + 1, VectorND<N>{});
        for (int32_t n : g.bdryNodes) {
            auto it = m_dirichletRegionOfNode.find((size_t)n);
            size_t r = it == m_dirichletRegionOfNode.end() ? 0 : it->second;
            for (size_t d = 0; d < N; ++d) forces[r][d] += f[(size_t)n][d];
        }
        return forces;
    }
    void reportRegionSurfaceForces(const VField &uNodes) const {
        auto forces = regionSurfaceForces(uNodes);
        for (size_t r = 0; r < forces.size(); ++r) {
            printf("region %zu surface force:", r);
            for (size_t d = 0; d < N; ++d) printf("\t%.10g", forces[r][d]);
            printf("\n");
        }
    }
    BBox<N> boundingBox() const { return BBox<N>(m_geometry().pos); }
    std::vector<int32_t> boundaryNodes() const { return m_geometry().bdryNodes; }

    // box-region boundary conditions, straight to the C ABI
    void applyDirichletBox(const std::array<Real, N> &mn, const std::array<Real, N> &mx, const std::array<Real, N> &value,
                           bool relative = false, int componentMask = (1 << N) - 1) {
        check(ctx(), mfh_bc_dirichlet_box(ctx(), mn.data(), mx.data(), relative, value.data(), componentMask));
    }
    void applyNeumannBox(const std::array<Real, N> &mn, const std::array<Real, N> &mx, const std::array<Real, N> &value,
                         int kind = MFH_NEUMANN_TRACTION, bool relative = false) {
        check(ctx(), mfh_bc_neumann_box(ctx(), mn.data(), mx.data(), relative, value.data(), kind));
    }

    void applyPeriodicConditions(Real epsilon = 1e-7) {          // :845-854
        int64_t nd = 0;
        check(ctx(), mfh_apply_periodic_conditions(ctx(), epsilon, &nd));
        m_numDoFs = (size_t)nd;
    }
    // new vertex positions on the same connectivity (:1279-1284): every setup phase is kept, elements are re-embedded
    void updateMeshNodePositions(const std::vector<std::array<Real, N>> &vertices) {
        check(ctx(), mfh_mesh_update_vertices(ctx(), &vertices[0][0]));
        m_geo.valid = false;
    }
    void removePeriodicConditions() {                            // :874-879
        check(ctx(), mfh_dof_map(ctx(), nullptr, 0));
        m_numDoFs = m_numNodes;
    }
    // node positions (vertex nodes first) and element volumes, as sim.mesh() exposes them
    VField nodes() const {
        VField p(m_numNodes);
        check(ctx(), mfh_mesh_get_node_positions(ctx(), &p[0][0]));
        return p;
    }
    std::vector<Real> elementVolumes() const {
        std::vector<Real> v((size_t)m_numElements);
        check(ctx(), mfh_mesh_get_elem_volumes(ctx(), v.data()));
        return v;
    }
    void setUsePinNoRigidTranslationConstraint(bool pin) { m_usePin = pin; }   // PeriodicHomogenization.hh:44-45
    void applyNoRigidMotionConstraint() { m_noRigidMotion = true; }            // m_useRigidMotionConstraint (:1214-1228)
    void removeNoRigidMotionConstraint() { m_noRigidMotion = false; }
    void setRigidMotionConstraintRHS(const std::vector<Real> &rhs) { m_rigidMotionRHS = rhs; }
    // block-Jacobi (MFH_PRECOND_BLOCK_JACOBI, default) or + rigid-body-mode coarse space (MFH_PRECOND_TWO_LEVEL)
    void setPreconditioner(int kind) { check(ctx(), mfh_set_preconditioner(ctx(), kind)); }

    VField neumannLoad() const {                                 // :703-717
        VField f(m_numDoFs);
        check(ctx(), mfh_neumann_load(ctx(), &f[0][0]));
        return f;
    }
    VField constantStrainLoad(const std::array<Real, N *(N + 1) / 2> &cstrainFlat) const {   // :551-562
        VField f(m_numDoFs);
        check(ctx(), mfh_constant_strain_load(ctx(), cstrainFlat.data(), &f[0][0]));
        return f;
    }

    // per-DoF field -> per-node field through the DoF map (identity unless periodic conditions are installed)   (:664-677)
    VField dofToNodeField(const VField &dofField) const {
        if (dofField.size() != m_numDoFs) throw std::runtime_error("dofToNodeField: one entry per DoF expected");
        std::vector<int32_t> dm(m_numNodes);
        int64_t nd = 0;
        check(ctx(), mfh_get_dof_map(ctx(), dm.data(), &nd));
        VField out(m_numNodes);
        for (size_t n = 0; n < m_numNodes; ++n) out[n] = dofField[(size_t)dm[n]];
        return out;
    }
    VField solve(const VField &f) const {                        // :479-487 + dofToNodeField :664-677
        VField u(m_numNodes);
        check(ctx(), mfh_sim_solve_constrained(ctx(), &f[0][0], m_flags(), m_rigidMotionRHS.data(), (int32_t)m_rigidMotionRHS.size(),
                                               &u[0][0], rtol, maxit, &info));
        return u;
    }
    // [solve(constantStrainLoad(e)) for e in cstrains] in one call -- the loop of solveCellProblems (PeriodicHomogenization.hh:47-53): under the
    // multigrid preconditioner the load vectors are formed on the device and the right-hand sides share the V-cycle's coarse levels
    std::vector<VField> solveConstantStrainLoads(const std::vector<std::array<Real, N *(N + 1) / 2>> &cstrains) const {
        std::vector<VField> w;
        if (!m_rigidMotionRHS.empty()) {          // a right-hand side for the constraint rows: one solve per strain
            for (const auto &e : cstrains) w.push_back(solve(constantStrainLoad(e)));
            return w;
        }
        const size_t ns = cstrains.size();
        std::vector<Real> flat(ns * m_numNodes * N);
        infos.assign(ns, mfh_solve_info{});
        check(ctx(), mfh_solve_cell_problems(ctx(), (int32_t)ns, &cstrains[0][0], m_flags(), flat.data(), rtol, maxit, infos.data()));
        if (ns) info = infos.back();
        for (size_t k = 0; k < ns; ++k) {
            VField u(m_numNodes);
            std::copy(flat.begin() + k * m_numNodes * N, flat.begin() + (k + 1) * m_numNodes * N, &u[0][0]);
            w.push_back(std::move(u));
        }
        return w;
    }
    VField solve() const {                                       // :657
        VField u(m_numNodes);
        check(ctx(), mfh_sim_solve_constrained(ctx(), nullptr, m_flags(), m_rigidMotionRHS.data(), (int32_t)m_rigidMotionRHS.size(),
                                               &u[0][0], rtol, maxit, &info));
        return u;
    }

    // upper triangle of K as summed triplets (m_assembleStiffnessMatrix + sumRepeated, :1408-1466)
    void m_assembleStiffnessMatrix(TripletMatrix &K) const {
        check(ctx(), mfh_assemble(ctx(), MFH_ASSEMBLE_GATHER));
        uint64_t nnz = 0;
        check(ctx(), mfh_export_upper_triplets(ctx(), nullptr, nullptr, nullptr, &nnz));
        std::vector<uint64_t> i(nnz), j(nnz);
        std::vector<Real> v(nnz);
        check(ctx(), mfh_export_upper_triplets(ctx(), i.data(), j.data(), v.data(), &nnz));
        K.m = K.n = N * m_numDoFs;
        K.nz.resize(nnz);
        for (uint64_t k = 0; k < nnz; ++k) K.nz[k] = Triplet{(size_t)i[k], (size_t)j[k], v[k]};
    }

    VField applyStiffnessMatrix(const VField &u) const {         // :801-823 (u per DoF)
        VField Ku(m_numDoFs);
        check(ctx(), mfh_apply_K(ctx(), &u[0][0], &Ku[0][0]));
        return Ku;
    }
    SMField averageStrainField(const VField &uNodes) const {     // :528-538
        SMField e((size_t)m_numElements);
        check(ctx(), mfh_average_strain(ctx(), &uNodes[0][0], &e[0][0]));
        return e;
    }
    // strainField / stressField (:511-526): nodal values of every element's strain interpolant, [nElem][1 | N+1][flatLen]
    std::vector<Real> strainField(const VField &uNodes, bool stress = false) const {
        std::vector<Real> out((size_t)m_numElements * (Deg == 1 ? 1 : N + 1) * (N * (N + 1) / 2));
        check(ctx(), mfh_strain_field(ctx(), &uNodes[0][0], stress ? 1 : 0, out.data()));
        return out;
    }
    std::vector<Real> stressField(const VField &uNodes) const { return strainField(uNodes, true); }
    // the same interpolant restricted to the boundary elements (restrictInterpolant, InterpolantRestriction.hh:29-66):
    // values at the boundary element's corners in its own vertex order, [nBdryElem][1 | N][flatLen]
    std::vector<Real> boundaryStrainField(const VField &uNodes, bool stress = false) const {
        int64_t nBE = 0;
        check(ctx(), mfh_mesh_sizes(ctx(), nullptr, nullptr, nullptr, &nBE, nullptr, nullptr, nullptr));
        std::vector<Real> out((size_t)nBE * (Deg == 1 ? 1 : N) * (N * (N + 1) / 2));
        check(ctx(), mfh_boundary_strain_field(ctx(), &uNodes[0][0], stress ? 1 : 0, out.data()));
        return out;
    }
    // sum_e vol_e C_e : (average strain_e(u) + cstrain), reduced on the device (the element loop of homogenizedElasticityTensor)
    std::array<Real, N *(N + 1) / 2> integratedStress(const VField &uNodes, const std::array<Real, N *(N + 1) / 2> *cstrainFlat = nullptr) const {
        std::array<Real, N *(N + 1) / 2> out{};
        check(ctx(), mfh_integrated_stress(ctx(), &uNodes[0][0], cstrainFlat ? cstrainFlat->data() : nullptr, out.data()));
        return out;
    }
    SMField averageStressField(const VField &uNodes) const {     // :539-549
        SMField s((size_t)m_numElements);
        check(ctx(), mfh_average_stress(ctx(), &uNodes[0][0], &s[0][0]));
        return s;
    }

    // ---- discrete shape derivatives, forward mode (:1297-1374); deltaP is a per-vertex field
    VField applyDeltaStiffnessMatrix(const VField &uNodes, const VField &deltaP) const {     // :1301-1328, per-DoF result
        VField f(m_numDoFs);
        check(ctx(), mfh_apply_delta_K(ctx(), &uNodes[0][0], &deltaP[0][0], &f[0][0]));
        return f;
    }
    VField deltaConstantStrainLoad(const std::array<Real, N *(N + 1) / 2> &cstrainFlat, const VField &deltaP) const {   // :1331-1348
        VField f(m_numDoFs);
        check(ctx(), mfh_delta_constant_strain_load(ctx(), cstrainFlat.data(), &deltaP[0][0], &f[0][0]));
        return f;
    }
    SMField deltaAverageStrainField(const VField &uNodes, const VField &deltaU, const VField &deltaP) const {   // :1364-1374
        SMField e((size_t)m_numElements);
        check(ctx(), mfh_delta_average_strain(ctx(), &uNodes[0][0], &deltaU[0][0], &deltaP[0][0], 0, &e[0][0]));
        return e;
    }

    double rtol = 1e-8;
    int maxit = 100000;
    mutable mfh_solve_info info{};
    mutable std::vector<mfh_solve_info> infos;    // one record per right-hand side of the last solveConstantStrainLoads

private:
    Context m_owner;
    int64_t m_numElements = 0;
    size_t m_numNodes = 0, m_numDoFs = 0;
    bool m_usePin = false, m_noRigidMotion = false;
    std::vector<Real> m_rigidMotionRHS;
    // host copies of the mesh tables the condition matching reads (fetched once, refreshed after a vertex update)
    struct Geometry {
        bool valid = false;
        VField pos;
        std::vector<int32_t> bdryNodes, beNodes;
        std::vector<Real> area, normal;
    };
    mutable Geometry m_geo;
    std::map<size_t, size_t> m_dirichletRegionOfNode;            // BoundaryNode::dirichletRegionIdx
    size_t m_dirichletRegionCount = 0;
    const Geometry &m_geometry() const {
        if (m_geo.valid) return m_geo;
        int64_t nBE = 0, nBN = 0;
        int32_t npbe = 0;
        check(ctx(), mfh_mesh_sizes(ctx(), nullptr, nullptr, nullptr, &nBE, &nBN, nullptr, &npbe));
        m_geo.pos = nodes();
        m_geo.bdryNodes.resize((size_t)nBN);
        m_geo.beNodes.resize((size_t)nBE * (size_t)npbe);
        m_geo.area.resize((size_t)nBE);
        m_geo.normal.resize((size_t)nBE * N);
        check(ctx(), mfh_mesh_get_boundary_nodes(ctx(), m_geo.bdryNodes.data()));
        check(ctx(), mfh_mesh_get_boundary_elem_nodes(ctx(), m_geo.beNodes.data()));
        check(ctx(), mfh_mesh_get_boundary_elem_geometry(ctx(), m_geo.area.data(), m_geo.normal.data()));
        m_geo.valid = true;
        return m_geo;
    }
    int32_t m_flags() const { return (m_usePin ? MFH_SOLVE_PIN : 0) | (m_noRigidMotion ? MFH_SOLVE_NO_RIGID_MOTION : 0); }
};

} // namespace LinearElasticity

// Scalar operators on the same kernels (1x1 blocks): Laplacian::construct (Laplacian.hh:97-104),
// MassMatrix::construct (MassMatrix.hh:103-128), PoissonMesh (Poisson.hh:55-132).
template <size_t N, size_t Deg>
class PoissonMesh {
public:
    PoissonMesh(const std::vector<std::array<int32_t, N + 1>> &elems, const std::vector<std::array<Real, N>> &vertices, int device = 0)
        : m_owner(device) {
        check(ctx(), mfh_mesh_build(ctx(), (int32_t)N, (int32_t)Deg, (int64_t)elems.size(), (int64_t)vertices.size(),
                                    &elems[0][0], &vertices[0][0]));
        int64_t nn = 0;
        check(ctx(), mfh_mesh_sizes(ctx(), &m_numElements, &nn, nullptr, nullptr, nullptr, nullptr, nullptr));
        m_numNodes = (size_t)nn;
        check(ctx(), mfh_set_operator(ctx(), MFH_OP_LAPLACIAN));
    }
    mfh_ctx *ctx() const { return m_owner.get(); }
    size_t numNodes() const { return m_numNodes; }
    size_t numElements() const { return (size_t)m_numElements; }

    // one DirichletCondition of applyBoundaryConditions (Poisson.hh:69-84): boundary nodes in the inclusive box
    void applyDirichletBox(const std::array<Real, N> &mn, const std::array<Real, N> &mx, Real value, bool relative = false) {
        std::array<Real, N> v{};
        v[0] = value;
        check(ctx(), mfh_set_operator(ctx(), MFH_OP_LAPLACIAN));
        check(ctx(), mfh_bc_dirichlet_box(ctx(), mn.data(), mx.data(), relative, v.data(), 1));
    }
    void solve(std::vector<Real> &x) {                                  // Poisson.hh:91-117
        x.resize(m_numNodes);
        check(ctx(), mfh_set_operator(ctx(), MFH_OP_LAPLACIAN));
        check(ctx(), mfh_sim_solve(ctx(), nullptr, 0, x.data(), rtol, maxit, &info));
    }
    std::vector<std::array<Real, N>> gradUAverage(const std::vector<Real> &u) const {   // :121-131
        std::vector<std::array<Real, N>> g((size_t)m_numElements);
        check(ctx(), mfh_average_gradient(ctx(), u.data(), &g[0][0]));
        return g;
    }
    // upper triangles after sumRepeated
    TripletMatrix laplacian() { return m_operator(MFH_OP_LAPLACIAN); }
    TripletMatrix massMatrix() { return m_operator(MFH_OP_MASS); }

    double rtol = 1e-10;
    int maxit = 100000;
    mfh_solve_info info{};

private:
    TripletMatrix m_operator(int op) {
        check(ctx(), mfh_set_operator(ctx(), op));
        check(ctx(), mfh_assemble(ctx(), MFH_ASSEMBLE_GATHER));
        uint64_t nnz = 0;
        check(ctx(), mfh_export_upper_triplets(ctx(), nullptr, nullptr, nullptr, &nnz));
        std::vector<uint64_t> i(nnz), j(nnz);
        std::vector<Real> v(nnz);
        check(ctx(), mfh_export_upper_triplets(ctx(), i.data(), j.data(), v.data(), &nnz));
        TripletMatrix K;
        K.m = K.n = m_numNodes;
        K.nz.resize(nnz);
        for (uint64_t k = 0; k < nnz; ++k) K.nz[k] = Triplet{(size_t)i[k], (size_t)j[k], v[k]};
        return K;
    }
    Context m_owner;
    int64_t m_numElements = 0;
    size_t m_numNodes = 0;
};

// SPSDSystem<Real>(K) for a caller-supplied SPD matrix (SparseMatrices.hh:2332-2348): owns its context.
class GenericSPSDSystem {
public:
    explicit GenericSPSDSystem(const TripletMatrix &K, int device = 0) : m_owner(device), m_n(K.m) {
        if (K.m != K.n) throw std::runtime_error("K must be square");
        std::vector<uint64_t> i, j;
        std::vector<Real> v;
        for (const auto &t : K.nz)
            if (t.i <= t.j) { i.push_back(t.i); j.push_back(t.j); v.push_back(t.v); }   // setUpperTriangle (:2337)
        check(m_owner.get(), mfh_matrix_set_upper_triplets(m_owner.get(), (int64_t)K.m, (int64_t)v.size(), i.data(), j.data(), v.data()));
    }
    void fixVariables(const std::vector<size_t> &fixedVars, const std::vector<Real> &fixedVarValues = std::vector<Real>()) {
        std::vector<int64_t> v(fixedVars.begin(), fixedVars.end());
        check(m_owner.get(), mfh_fix_variables(m_owner.get(), (int64_t)v.size(), v.data(), fixedVarValues.empty() ? nullptr : fixedVarValues.data()));
    }
    void solve(const std::vector<Real> &b, std::vector<Real> &x) {
        if (b.size() != m_n) throw std::runtime_error("Bad RHS");
        x.resize(m_n);
        check(m_owner.get(), mfh_solve(m_owner.get(), 1, b.data(), x.data(), rtol, maxit, &info));
    }
    double rtol = 1e-10;
    int maxit = 100000;
    mfh_solve_info info{};

private:
    Context m_owner;
    size_t m_n;
};
} // namespace MeshFEMHip

#endif /* end of include guard: MESHFEMHIP_LINEARELASTICITY_HH */
