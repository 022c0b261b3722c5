// MeshFEMHip/PeriodicHomogenization.hh -- header-only C++ facade over the C ABI for the second caller of the hot path,
// keeping the reference's free-function names so that a PeriodicHomogenization_cli-style driver only changes its includes:
//
//     reference (src/lib/MeshFEM/PeriodicHomogenization.hh)                this header (namespace MeshFEMHip::PeriodicHomogenization)
//       solveCellProblems(w_ij, sim, cellEpsilon)               :34-54       same
//       homogenizedElasticityTensor(w_ij, sim, baseCellVolume)  :72-103      same (stress-like form)
//       homogenizedElasticityTensorDisplacementForm(...)        :146-186     same (boundary-integral form, constant base tensor)
//       deltaFluctuationDisplacements(sim, w, delta_p)          :527-544     same
//       deltaHomogenizedElasticityTensor(sim, w, delta_p)       :492-514     exact discrete derivative, volume form (:484-491)
//       homogenizedElasticityTensorGradient(w, sim)             :226-288     same, packed [bdryElem][node][flatLen][flatLen]
//       (boundary integral of the above against n . delta_p)    :492-514     deltaHomogenizedElasticityTensorBoundaryForm
//       homogenizedElasticityTensorDiscreteDifferential(w, sim) :372-480     same, packed [pair][vertex][component]
//     reference (OrthotropicHomogenization.hh)                              namespace ...::Orthotropic
//       solveCellProblems(w_ij, sim, cellEpsilon)               :44-153      same (one assembled operator, 1 + flatLen - N masks)
//       homogenizedTensorFromOrthoCellQuantity(EhO)             :183-198     same
//       homogenizedElasticityTensor[DisplacementForm](w, sim)   :200-216     same
//
// Tensors are flattened flatLen x flatLen matrices (Voigt order xx,yy,zz,yz,xz,xy; tensor shear entries), the reference's
// ElasticityTensor::D. Errors of the C ABI are rethrown as std::runtime_error.
#ifndef MESHFEMHIP_PERIODICHOMOGENIZATION_HH
#define MESHFEMHIP_PERIODICHOMOGENIZATION_HH

#include <cmath>

#include "LinearElasticity.hh"

namespace MeshFEMHip {
namespace PeriodicHomogenization {

constexpr size_t flatLen(size_t N) { return N * (N + 1) / 2; }

template <size_t N> struct ETensor {
    static constexpr size_t FL = N * (N + 1) / 2;
    std::array<std::array<Real, FL>, FL> D{};
    Real &operator()(size_t i, size_t j) { return D[i][j]; }
    Real operator()(size_t i, size_t j) const { return D[i][j]; }
};

// SymmetricMatrix CanonicalBasis(k) flattened: 1 on the diagonal entries, 1/2 for shear (SymmetricMatrix.hh:405-413)
template <size_t N> std::array<Real, N *(N + 1) / 2> canonicalStrain(size_t k, Real scale = 1.0) {
    std::array<Real, N *(N + 1) / 2> e{};
    e[k] = scale * (k < N ? 1.0 : 0.5);
    return e;
}

namespace detail {
template <size_t N, size_t Deg> Real cellVolume(const LinearElasticity::Simulator<N, Deg> &sim, Real baseCellVolume) {
    if (baseCellVolume != 0.0) return baseCellVolume;
    auto p = sim.nodes();
    Real vol = 1.0;
    for (size_t a = 0; a < N; ++a) {
        Real lo = p[0][a], hi = p[0][a];
        for (const auto &x : p) { lo = std::min(lo, x[a]); hi = std::max(hi, x[a]); }
        vol *= hi - lo;
    }
    return vol;
}
// flattened index -> (row, col) of the symmetric matrix (Flattening.hh:62-83)
template <size_t N> void unflatten(size_t k, size_t &i, size_t &j) {
    if (k < N) { i = j = k; return; }
    if (N == 2) { i = 0; j = 1; return; }
    if (k == 3) { i = 1; j = 2; } else if (k == 4) { i = 0; j = 2; } else { i = 0; j = 1; }
}
template <size_t N, size_t Deg>
std::vector<Real> stackFields(const LinearElasticity::Simulator<N, Deg> &sim, const std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w) {
    if (w.size() != flatLen(N)) throw std::runtime_error("need one fluctuation displacement per canonical strain");
    std::vector<Real> out;
    out.reserve(w.size() * sim.numNodes() * N);
    for (const auto &f : w) {
        if (f.size() != sim.numNodes()) throw std::runtime_error("fluctuation displacements are per-node fields");
        for (const auto &x : f) out.insert(out.end(), x.begin(), x.end());
    }
    return out;
}
// end points of the boundary element's local edges (FEMMesh.inl:43-58)
static const size_t es[3] = {0, 1, 2}, et[3] = {1, 2, 0};
}   // namespace detail

// Solve the cell problems -div E : [strain(w^ij) + e^ij] = 0, w^ij periodic, pinned (PeriodicHomogenization.hh:34-54).
// The matrix is assembled once; every canonical strain is one PCG solve with rhs = constantStrainLoad(-e_ij).
template <size_t N, size_t Deg>
void solveCellProblems(std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w_ij, LinearElasticity::Simulator<N, Deg> &sim,
                       Real cellEpsilon = 1e-7) {
    sim.applyPeriodicConditions(cellEpsilon);
    sim.applyNoRigidMotionConstraint();
    sim.setUsePinNoRigidTranslationConstraint(true);
    // one system, flatLen right-hand sides (the reference factors once and back-substitutes per load): handed over together, so that the load
    // vectors are formed on the device and the preconditioner's coarse levels serve all of them at once (mfh_solve_cell_problems)
    std::vector<std::array<Real, flatLen(N)>> strains;
    for (size_t k = 0; k < flatLen(N); ++k) strains.push_back(canonicalStrain<N>(k, -1.0));
    w_ij = sim.solveConstantStrainLoads(strains);
}

// Stress-like form: Eh.DRow(i) = 1/|Y| sum_e vol_e [E_e : avg strain(w_i) + E_e.DRow(i)]  (:72-103). E_e.DRow(i) is the
// average stress of the linear displacement with constant strain e_i, so no per-element tensor is read back.
template <size_t N, size_t Deg>
ETensor<N> homogenizedElasticityTensor(const std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w_ij,
                                       const LinearElasticity::Simulator<N, Deg> &sim, Real baseCellVolume = 0.0) {
    constexpr size_t FL = flatLen(N);
    const Real cell = detail::cellVolume(sim, baseCellVolume);
    ETensor<N> Eh;
    for (size_t i = 0; i < FL; ++i) {
        // Eh.DRow(i) = 1/|Y| sum_e vol_e E_e : (avg strain(w_i) + e_i): one device reduction (mfh_integrated_stress)
        std::array<Real, FL> ei{};
        ei[i] = i < N ? 1.0 : 0.5;               // CanonicalBasis(i), tensor shear entries
        const auto row = sim.integratedStress(w_ij.at(i), &ei);
        for (size_t j = 0; j < FL; ++j) Eh.D[i][j] = row[j] / cell;
    }
    return Eh;
}

// Energy form on the device: Ch_ijkl = 1/|Y| int (e^ij + eps(w^ij)) : C : (e^kl + eps(w^kl)) dV (one reduction per entry);
// equal to the other forms at the cell-problem solutions.
template <size_t N, size_t Deg>
ETensor<N> homogenizedElasticityTensorEnergyForm(const std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w_ij,
                                                 const LinearElasticity::Simulator<N, Deg> &sim, Real baseCellVolume = 0.0) {
    constexpr size_t FL = flatLen(N);
    const Real cell = detail::cellVolume(sim, baseCellVolume);
    const auto w = detail::stackFields(sim, w_ij);
    std::array<Real, FL * FL> out{};
    check(sim.ctx(), mfh_mutual_energies(sim.ctx(), w.data(), nullptr, out.data()));
    ETensor<N> Eh;
    for (size_t i = 0; i < FL; ++i)
        for (size_t j = 0; j < FL; ++j) Eh.D[i][j] = out[i * FL + j] / cell;
    return Eh;
}

// Displacement (boundary-integral) form, the one PeriodicHomogenization_cli and the Python binding print (:146-186):
// Eh.DRow(i) = 1/|Y| [E : sum_be sym(int_be w_i (x) n) + E vol(omega)], constant base tensor `EBase` (flattened).
template <size_t N, size_t Deg>
ETensor<N> homogenizedElasticityTensorDisplacementForm(const std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w_ij,
                                                       const LinearElasticity::Simulator<N, Deg> &sim, Real baseCellVolume = 0.0) {
    constexpr size_t FL = flatLen(N);
    const Real cell = detail::cellVolume(sim, baseCellVolume);
    mfh_ctx *c = sim.ctx();
    int64_t nBE = 0, nBN = 0;
    check(c, mfh_mesh_sizes(c, nullptr, nullptr, nullptr, &nBE, &nBN, nullptr, nullptr));
    const size_t npbe = (Deg == 1) ? N : (N == 2 ? 3 : 6);
    std::vector<int32_t> ben((size_t)nBE * npbe);
    std::vector<Real> area((size_t)nBE), nrm((size_t)nBE * N);
    check(c, mfh_mesh_get_boundary_elem_nodes(c, ben.data()));
    check(c, mfh_mesh_get_boundary_elem_geometry(c, area.data(), nrm.data()));
    // integrals of the boundary element's nodal shape functions over a unit simplex (Functions.hh:246-318)
    std::array<Real, 6> wt{};
    if (Deg == 1) for (size_t k = 0; k < npbe; ++k) wt[k] = 1.0 / npbe;
    else if (N == 2) { wt[0] = wt[1] = 1.0 / 6.0; wt[2] = 4.0 / 6.0; }
    else { wt[3] = wt[4] = wt[5] = 1.0 / 3.0; }
    ETensor<N> EBase;
    {
        std::array<Real, FL * FL> d{};
        check(c, mfh_material_get(c, 0, d.data()));
        for (size_t i = 0; i < FL; ++i)
            for (size_t j = 0; j < FL; ++j) EBase.D[i][j] = d[i * FL + j];
    }
    Real volume = 0;
    for (Real v : sim.elementVolumes()) volume += v;
    ETensor<N> Eh;
    for (size_t i = 0; i < FL; ++i) {
        std::array<Real, FL> nw{};
        for (int64_t b = 0; b < nBE; ++b) {
            std::array<Real, N> wint{};
            for (size_t k = 0; k < npbe; ++k)
                for (size_t a = 0; a < N; ++a) wint[a] += wt[k] * w_ij.at(i)[(size_t)ben[(size_t)b * npbe + k]][a] * area[(size_t)b];
            for (size_t q = 0; q < FL; ++q) {
                size_t a, bb;
                detail::unflatten<N>(q, a, bb);
                nw[q] += 0.5 * (wint[a] * nrm[(size_t)b * N + bb] + wint[bb] * nrm[(size_t)b * N + a]);
            }
        }
        for (size_t r = 0; r < FL; ++r) {   // doubleContract: shear entries doubled (ElasticityTensor.hh:437-449)
            Real v = 0;
            for (size_t q = 0; q < FL; ++q) v += EBase.D[r][q] * nw[q] * (q < N ? 1.0 : 2.0);
            Eh.D[i][r] = (v + EBase.D[i][r] * volume) / cell;
        }
    }
    return Eh;
}

// Change of the cell-problem solutions under the vertex perturbation delta_p (:527-544)
template <size_t N, size_t Deg>
std::vector<typename LinearElasticity::Simulator<N, Deg>::VField>
deltaFluctuationDisplacements(const LinearElasticity::Simulator<N, Deg> &sim, const std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w,
                              const typename LinearElasticity::Simulator<N, Deg>::VField &delta_p) {
    std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> dw;
    for (size_t k = 0; k < w.size(); ++k) {
        auto rhs = sim.deltaConstantStrainLoad(canonicalStrain<N>(k, -1.0), delta_p);
        const auto dKw = sim.applyDeltaStiffnessMatrix(w[k], delta_p);
        for (size_t n = 0; n < rhs.size(); ++n)
            for (size_t a = 0; a < N; ++a) rhs[n][a] -= dKw[n][a];
        dw.push_back(sim.solve(rhs));
    }
    return dw;
}

// Change of Ch under delta_p with |Y| fixed (:492-514), evaluated in the volume form quoted at :484-491
template <size_t N, size_t Deg>
ETensor<N> deltaHomogenizedElasticityTensor(const LinearElasticity::Simulator<N, Deg> &sim,
                                            const std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w,
                                            const typename LinearElasticity::Simulator<N, Deg>::VField &delta_p, Real baseCellVolume = 0.0) {
    constexpr size_t FL = flatLen(N);
    const Real cell = detail::cellVolume(sim, baseCellVolume);
    const auto ws = detail::stackFields(sim, w);
    std::array<Real, FL * FL> out{};
    check(sim.ctx(), mfh_mutual_energies(sim.ctx(), ws.data(), &delta_p[0][0], out.data()));
    ETensor<N> d;
    for (size_t i = 0; i < FL; ++i)
        for (size_t j = 0; j < FL; ++j) d.D[i][j] = out[i * FL + j] / cell;
    return d;
}

// Steepest-ascent normal velocity of every component of Ch (:226-288): per boundary element the nodal values of the degree
// 2 (Deg - 1) interpolant G_ijkl = 1/|bbox| (e_ij + eps(w_ij)) : E : (e_kl + eps(w_kl)), zero on the periodic boundary.
// Packed [bdryElem][1 | nodesPerBdryElem][flatLen][flatLen]; boundary edge nodes are (0,1),(1,2),(2,0).
template <size_t N, size_t Deg>
std::vector<Real> homogenizedElasticityTensorGradient(const std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w,
                                                      const LinearElasticity::Simulator<N, Deg> &sim) {
    constexpr size_t FL = flatLen(N), NQ = Deg == 1 ? 1 : N, NE = Deg == 1 ? 0 : (N == 2 ? 1 : 3), NN = NQ + NE;
    if (w.size() != FL) throw std::runtime_error("need one fluctuation displacement per canonical strain");
    const Real bbox = detail::cellVolume(sim, 0.0);
    const auto pos = sim.nodes();
    int64_t nBE = 0;
    check(sim.ctx(), mfh_mesh_sizes(sim.ctx(), nullptr, nullptr, nullptr, &nBE, nullptr, nullptr, nullptr));
    std::vector<uint8_t> internal((size_t)nBE);
    check(sim.ctx(), mfh_mesh_get_boundary_elem_internal(sim.ctx(), internal.data()));
    // G = e + eps(w) and S = E : G at the boundary corners, then at the edge midpoints (both are linear on the element)
    std::vector<std::vector<Real>> G(FL), S(FL);
    for (size_t k = 0; k < FL; ++k) {
        size_t a, b;
        detail::unflatten<N>(k, a, b);
        auto u = w[k];
        const Real sc = (a == b) ? 1.0 : 0.5;
        for (size_t n = 0; n < u.size(); ++n) {   // u_lin = e_k x
            u[n][a] += sc * pos[n][b];
            if (a != b) u[n][b] += sc * pos[n][a];
        }
        G[k] = sim.boundaryStrainField(u, false);
        S[k] = sim.boundaryStrainField(u, true);
    }
    using detail::es; using detail::et;
    auto at = [&](const std::vector<Real> &X, size_t be, size_t n, size_t c) {
        if (n < NQ) return X[(be * NQ + n) * FL + c];
        return 0.5 * (X[(be * NQ + es[n - NQ]) * FL + c] + X[(be * NQ + et[n - NQ]) * FL + c]);
    };
    std::vector<Real> out((size_t)nBE * NN * FL * FL, 0.0);
    for (size_t be = 0; be < (size_t)nBE; ++be) {
        if (internal[be]) continue;
        for (size_t n = 0; n < NN; ++n)
            for (size_t i = 0; i < FL; ++i)
                for (size_t k = i; k < FL; ++k) {
                    Real v = 0.0;
                    for (size_t c = 0; c < FL; ++c) v += (c < N ? 1.0 : 2.0) * at(S[i], be, n, c) * at(G[k], be, n, c);
                    out[((be * NN + n) * FL + i) * FL + k] = out[((be * NN + n) * FL + k) * FL + i] = v / bbox;
                }
    }
    return out;
}

// deltaHomogenizedElasticityTensor exactly as the reference evaluates it (:492-514): the linear normal velocity
// n . delta_p of every boundary element integrated against homogenizedElasticityTensorGradient (continuous form).
template <size_t N, size_t Deg>
ETensor<N> deltaHomogenizedElasticityTensorBoundaryForm(const LinearElasticity::Simulator<N, Deg> &sim,
                                                        const std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w,
                                                        const typename LinearElasticity::Simulator<N, Deg>::VField &delta_p) {
    constexpr size_t FL = flatLen(N), NQ = Deg == 1 ? 1 : N, NE = Deg == 1 ? 0 : (N == 2 ? 1 : 3), NN = NQ + NE;
    const auto sd = homogenizedElasticityTensorGradient(w, sim);
    int64_t nBE = 0;
    int32_t npbe = 0;
    check(sim.ctx(), mfh_mesh_sizes(sim.ctx(), nullptr, nullptr, nullptr, &nBE, nullptr, nullptr, &npbe));
    std::vector<int32_t> ben((size_t)nBE * npbe);
    std::vector<Real> area((size_t)nBE), nrm((size_t)nBE * N);
    check(sim.ctx(), mfh_mesh_get_boundary_elem_nodes(sim.ctx(), ben.data()));
    check(sim.ctx(), mfh_mesh_get_boundary_elem_geometry(sim.ctx(), area.data(), nrm.data()));
    using detail::es; using detail::et;
    // W[a][n] = int lambda_a phi_n over the unit-volume boundary simplex (exact)
    Real W[N][NN];
    for (size_t a = 0; a < N; ++a)
        for (size_t n = 0; n < NN; ++n) {
            if (Deg == 1) W[a][n] = 1.0 / N;
            else if (N == 2) W[a][n] = n == 2 ? 1.0 / 3.0 : (n == a ? 1.0 / 6.0 : 0.0);
            else if (n < 3) W[a][n] = n == a ? 1.0 / 30.0 : -1.0 / 60.0;
            else W[a][n] = (es[n - 3] == a || et[n - 3] == a) ? 2.0 / 15.0 : 1.0 / 15.0;
        }
    ETensor<N> d;
    for (size_t be = 0; be < (size_t)nBE; ++be)
        for (size_t a = 0; a < N; ++a) {
            Real nsv = 0.0;
            for (size_t c = 0; c < N; ++c) nsv += nrm[be * N + c] * delta_p[(size_t)ben[be * npbe + a]][c];
            for (size_t n = 0; n < NN; ++n) {
                const Real f = area[be] * nsv * W[a][n];
                for (size_t i = 0; i < FL; ++i)
                    for (size_t k = 0; k < FL; ++k) d.D[i][k] += f * sd[((be * NN + n) * FL + i) * FL + k];
            }
        }
    return d;
}

// The per-vertex one-form dCh (:372-480), packed [pair ij <= kl, row-major][vertex][component]
template <size_t N, size_t Deg>
std::vector<Real> homogenizedElasticityTensorDiscreteDifferential(const std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w,
                                                                  const LinearElasticity::Simulator<N, Deg> &sim, Real baseCellVolume = 0.0) {
    constexpr size_t FL = flatLen(N);
    const Real cell = detail::cellVolume(sim, baseCellVolume);
    const auto ws = detail::stackFields(sim, w);
    int64_t nVert = 0;
    check(sim.ctx(), mfh_mesh_sizes(sim.ctx(), nullptr, nullptr, &nVert, nullptr, nullptr, nullptr, nullptr));
    std::vector<Real> out(FL * (FL + 1) / 2 * (size_t)nVert * N);
    check(sim.ctx(), mfh_mutual_energy_differential(sim.ctx(), ws.data(), out.data()));
    for (auto &x : out) x /= cell;
    return out;
}

namespace Orthotropic {

// sign of probe ij's fluctuation under the reflection with bit mask r (OrthotropicHomogenization.hh:161-174)
template <size_t N> Real fluctuationDisplacementSign(size_t ij, size_t r) {
    if (ij < N) return 1.0;
    size_t count = 0;
    for (size_t b = 0; b < N; ++b)
        if (((r >> b) & 1) && !(N == 3 && b == ij - N)) ++count;
    return count == 1 ? -1.0 : 1.0;
}

template <size_t N> ETensor<N> homogenizedTensorFromOrthoCellQuantity(const ETensor<N> &EhO) {   // :183-198
    constexpr size_t FL = flatLen(N);
    ETensor<N> Eh;
    for (size_t r = 0; r < (size_t(1) << N); ++r)
        for (size_t kl = 0; kl < FL; ++kl)
            for (size_t ij = 0; ij <= kl; ++ij)
                Eh.D[ij][kl] += fluctuationDisplacementSign<N>(ij, r) * fluctuationDisplacementSign<N>(kl, r) * EhO.D[ij][kl] / Real(size_t(1) << N);
    for (size_t kl = 0; kl < FL; ++kl)
        for (size_t ij = 0; ij < kl; ++ij) Eh.D[kl][ij] = Eh.D[ij][kl];
    return Eh;
}

// Cell problems on the orthotropic base cell (:44-153): no periodicity, no rigid-motion rows; components are fixed on the
// symmetry planes. The reference builds 1 + (flatLen - N) SPSDSystems holding copies of K; here only the mask changes.
template <size_t N, size_t Deg>
void solveCellProblems(std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w_ij, LinearElasticity::Simulator<N, Deg> &sim,
                       Real cellEpsilon = 1e-7) {
    using VField = typename LinearElasticity::Simulator<N, Deg>::VField;
    constexpr size_t FL = flatLen(N);
    sim.removePeriodicConditions();
    sim.removeNoRigidMotionConstraint();
    mfh_ctx *c = sim.ctx();
    check(c, mfh_assemble(c, MFH_ASSEMBLE_GATHER));
    const auto pos = sim.nodes();
    std::array<Real, N> lo = pos[0], hi = pos[0];
    for (const auto &x : pos)
        for (size_t a = 0; a < N; ++a) { lo[a] = std::min(lo[a], x[a]); hi[a] = std::max(hi[a], x[a]); }
    auto onFace = [&](size_t n, size_t a) { return std::fabs(pos[n][a] - lo[a]) <= cellEpsilon || std::fabs(pos[n][a] - hi[a]) <= cellEpsilon; };
    std::vector<VField> loads;
    for (size_t k = 0; k < FL; ++k) loads.push_back(sim.constantStrainLoad(canonicalStrain<N>(k, -1.0)));
    w_ij.assign(FL, VField());
    SPSDSystem system(c);
    system.rtol = sim.rtol; system.maxit = sim.maxit;
    for (size_t si = 0; si < 1 + FL - N; ++si) {
        std::vector<char> fix(N * pos.size(), 0);
        for (size_t n = 0; n < pos.size(); ++n)
            for (size_t a = 0; a < N; ++a) {
                if (!onFace(n, a)) continue;
                if (si == 0) fix[N * n + a] = 1;                               // stretch probes: w_a = 0 on the plane with normal e_a
                else if (N == 3) {
                    const size_t s = si - 1;
                    fix[N * n + s] = 1;                                        // perpendicular to the shear plane
                    if (a != s) fix[N * n + (N - (a + s))] = 1;                // neither a nor s
                } else fix[N * n + (a == 0 ? 1 : 0)] = 1;
            }
        std::vector<size_t> vars;
        for (size_t v = 0; v < fix.size(); ++v) if (fix[v]) vars.push_back(v);
        check(c, mfh_clear_fixed(c));
        system.fixVariables(vars, std::vector<Real>(vars.size(), 0.0));
        const size_t first = si == 0 ? 0 : N + si - 1, last = si == 0 ? N : N + si;
        for (size_t k = first; k < last; ++k) {
            std::vector<Real> f(N * pos.size()), u;
            for (size_t n = 0; n < pos.size(); ++n)
                for (size_t a = 0; a < N; ++a) f[N * n + a] = loads[k][n][a];
            system.solve(f, u);
            w_ij[k].resize(pos.size());
            for (size_t n = 0; n < pos.size(); ++n)
                for (size_t a = 0; a < N; ++a) w_ij[k][n][a] = u[N * n + a];
        }
    }
    check(c, mfh_clear_fixed(c));
}

template <size_t N, size_t Deg>
ETensor<N> homogenizedElasticityTensorDisplacementForm(const std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w_ij,
                                                       const LinearElasticity::Simulator<N, Deg> &sim, Real baseCellVolume = 0.0) {
    return homogenizedTensorFromOrthoCellQuantity<N>(PeriodicHomogenization::homogenizedElasticityTensorDisplacementForm(w_ij, sim, baseCellVolume));
}

template <size_t N, size_t Deg>
ETensor<N> homogenizedElasticityTensor(const std::vector<typename LinearElasticity::Simulator<N, Deg>::VField> &w_ij,
                                       const LinearElasticity::Simulator<N, Deg> &sim, Real baseCellVolume = 0.0) {
    return homogenizedTensorFromOrthoCellQuantity<N>(PeriodicHomogenization::homogenizedElasticityTensor(w_ij, sim, baseCellVolume));
}

}   // namespace Orthotropic
}   // namespace PeriodicHomogenization
}   // namespace MeshFEMHip

#endif
