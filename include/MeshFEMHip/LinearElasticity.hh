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

    // box-region boundary conditions (applyBoundaryConditions, :881-1027)
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

    VField solve(const VField &f) const {                        // :479-487 + dofToNodeField :664-677
        VField u(m_numNodes);
        check(ctx(), mfh_sim_solve_constrained(ctx(), &f[0][0], m_flags(), m_rigidMotionRHS.data(), (int32_t)m_rigidMotionRHS.size(),
                                               &u[0][0], rtol, maxit, &info));
        return u;
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

private:
    Context m_owner;
    int64_t m_numElements = 0;
    size_t m_numNodes = 0, m_numDoFs = 0;
    bool m_usePin = false, m_noRigidMotion = false;
    std::vector<Real> m_rigidMotionRHS;
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
