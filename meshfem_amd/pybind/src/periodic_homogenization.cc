// `periodic_homogenization` (== /root/reference/src/python_bindings/periodic_homogenization.cc:36-183): homogenize and probe on the
// MI355X path -- the cell problems are PCG solves on the device behind the C++ facade (include/MeshFEMHip/PeriodicHomogenization.hh:
// solveCellProblems, homogenizedElasticityTensorDisplacementForm, the Orthotropic variants), the homogenized tensor is the
// DISPLACEMENT form like the binding, fluctuations are centred by default.
#include "common.hh"

namespace {

namespace PH = MeshFEMHip::PeriodicHomogenization;
namespace LE = MeshFEMHip::LinearElasticity;

struct HomogenizationResult {
    py::object Ch;                 // tensors.ElasticityTensor{2,3}D
    std::vector<ArrD> w_ij;        // numNodes x N each
    std::vector<ArrD> strain_w_ij; // numElements x flatLen each
};

template <size_t N, size_t Deg>
HomogenizationResult run(const py::object &mesh, const ETensor<N> &Cbase, bool orthotropicCell, const std::string &manualPeriodicVerticesFile,
                         bool centerFluctuationDisplacements, bool ignorePeriodicMismatch, int device, int preconditioner, double rtol) {
    constexpr size_t FL = flatLenOf(N);
    if (!manualPeriodicVerticesFile.empty())
        throw std::runtime_error("manualPeriodicVerticesFile: use meshfem_amd.homogenization.solve_cell_problems (node-pair files are read there)");
    const ArrD V = mesh.attr("vertices")().cast<ArrD>();
    const ArrI F = mesh.attr("elements")().cast<ArrI>();
    std::vector<std::array<Real, N>> verts((size_t)V.shape(0));
    std::vector<std::array<int32_t, N + 1>> elems((size_t)F.shape(0));
    auto v = V.unchecked<2>();
    auto f = F.unchecked<2>();
    for (size_t i = 0; i < verts.size(); ++i) for (size_t a = 0; a < N; ++a) verts[i][a] = v(i, a);
    for (size_t e = 0; e < elems.size(); ++e) for (size_t k = 0; k <= N; ++k) elems[e][k] = (int32_t)f(e, k);
    LE::Simulator<N, Deg> sim(elems, verts, device);
    sim.rtol = rtol;
    sim.setPreconditioner(preconditioner);
    sim.setMaterialTensor(Cbase.flat());
    if (ignorePeriodicMismatch) MeshFEMHip::check(sim.ctx(), mfh_set_option(sim.ctx(), "periodic_ignore_mismatch", 1.0));
    std::vector<typename LE::Simulator<N, Deg>::VField> w;
    PH::ETensor<N> Ch;                                       // (the base tensor is read back from the context's material)
    if (orthotropicCell) {                                   // periodic_homogenization.cc:53-56
        PH::Orthotropic::solveCellProblems(w, sim);
        Ch = PH::Orthotropic::homogenizedElasticityTensorDisplacementForm(w, sim);
    } else {
        PH::solveCellProblems(w, sim);
        Ch = PH::homogenizedElasticityTensorDisplacementForm(w, sim);
    }
    if (centerFluctuationDisplacements)                      // :62-70
        for (auto &wk : w) {
            std::array<Real, N> mean{};
            for (const auto &x : wk) for (size_t a = 0; a < N; ++a) mean[a] += x[a];
            for (size_t a = 0; a < N; ++a) mean[a] /= (Real)wk.size();
            for (auto &x : wk) for (size_t a = 0; a < N; ++a) x[a] -= mean[a];
        }
    HomogenizationResult r;
    ETensor<N> out;
    for (size_t i = 0; i < FL; ++i) for (size_t j = 0; j < FL; ++j) out.D[i][j] = Ch.D[i][j];
    r.Ch = py::cast(out);
    for (const auto &wk : w) {
        ArrD a = make2d(wk.size(), N);
        auto o = a.mutable_unchecked<2>();
        for (size_t n = 0; n < wk.size(); ++n) for (size_t k = 0; k < N; ++k) o(n, k) = wk[n][k];
        r.w_ij.push_back(a);
        const auto s = sim.averageStrainField(wk);
        ArrD b = make2d(s.size(), FL);
        auto q = b.mutable_unchecked<2>();
        for (size_t e = 0; e < s.size(); ++e) for (size_t k = 0; k < FL; ++k) q(e, k) = s[e][k];
        r.strain_w_ij.push_back(b);
    }
    return r;
}

template <size_t N>
HomogenizationResult homogenizeN(const py::object &mesh, const ETensor<N> &Cbase, bool orthotropicCell, const std::string &file, bool center, bool ignoreMismatch,
                                 int device, int preconditioner, double rtol) {
    const size_t deg = mesh.attr("degree").cast<size_t>();
    if (mesh.attr("embeddingDimension").cast<size_t>() != N) throw std::runtime_error("mesh and base tensor have different dimensions");
    if (deg == 1) return run<N, 1>(mesh, Cbase, orthotropicCell, file, center, ignoreMismatch, device, preconditioner, rtol);
    if (deg == 2) return run<N, 2>(mesh, Cbase, orthotropicCell, file, center, ignoreMismatch, device, preconditioner, rtol);
    throw std::runtime_error("degree must be 1 or 2");
}

// probe (:92-150): u = E x + w with the face-average translation removed; strain = E + strain(w)
template <size_t N> py::tuple probeN(const py::object &mesh, const HomogenizationResult &hr, const std::array<Real, flatLenOf(N)> &ms) {
    constexpr size_t FL = flatLenOf(N);
    if (hr.w_ij.size() != FL) throw std::runtime_error("homogenization result and macro strain have different dimensions");
    const size_t nn = (size_t)hr.w_ij[0].shape(0), ne = (size_t)hr.strain_w_ij[0].shape(0);
    ArrD u = make2d(nn, N), su = make2d(ne, FL);
    auto U = u.mutable_unchecked<2>();
    auto S = su.mutable_unchecked<2>();
    for (size_t n = 0; n < nn; ++n) for (size_t a = 0; a < N; ++a) U(n, a) = 0.0;
    for (size_t e = 0; e < ne; ++e) for (size_t k = 0; k < FL; ++k) S(e, k) = 0.0;
    for (size_t i = 0; i < FL; ++i) {
        const Real c = (i < N ? 1.0 : 2.0) * ms[i];                      // shearDoubler
        auto w = hr.w_ij[i].unchecked<2>();
        auto s = hr.strain_w_ij[i].unchecked<2>();
        for (size_t n = 0; n < nn; ++n) for (size_t a = 0; a < N; ++a) U(n, a) += c * w(n, a);
        for (size_t e = 0; e < ne; ++e) for (size_t k = 0; k < FL; ++k) S(e, k) += c * s(e, k);
    }
    const ArrD nodes = mesh.attr("nodes")().cast<ArrD>();
    const ArrI bn = mesh.attr("boundaryNodes")().cast<ArrI>();
    auto P = nodes.unchecked<2>();
    std::array<Real, N> lo;
    for (size_t a = 0; a < N; ++a) lo[a] = 1e300;
    for (size_t n = 0; n < nn; ++n) for (size_t a = 0; a < N; ++a) lo[a] = std::min(lo[a], P(n, a));
    for (size_t d = 0; d < N; ++d) {                                    // face-average translation removal (:114-130)
        Real t = 0, cnt = 0;
        for (py::ssize_t k = 0; k < bn.size(); ++k) {
            const int64_t n = bn.at(k);
            if (std::fabs(P(n, d) - lo[d]) < 1e-9) { t += U(n, d); cnt += 1.0; }
        }
        t /= cnt;
        for (size_t n = 0; n < nn; ++n) U(n, d) -= t;
    }
    for (size_t n = 0; n < nn; ++n)                                       // u += macroStrain . x
        for (size_t a = 0; a < N; ++a) for (size_t b = 0; b < N; ++b) U(n, a) += ms[flattenIndices<N>(a, b)] * P(n, b);
    for (size_t e = 0; e < ne; ++e) for (size_t k = 0; k < FL; ++k) S(e, k) += ms[k];
    return py::make_tuple(u, su);
}

template <size_t N> std::array<Real, flatLenOf(N)> strainOf(const py::object &macroStrain) {
    std::array<Real, flatLenOf(N)> ms;
    if (py::isinstance<SMValue<N>>(macroStrain)) return macroStrain.cast<SMValue<N>>().flat;
    const ArrD a = macroStrain.cast<ArrD>();
    if (a.ndim() != 1 || (size_t)a.shape(0) != flatLenOf(N)) throw std::runtime_error("macroStrain: expected a SymmetricMatrix or its flattened values");
    for (size_t k = 0; k < flatLenOf(N); ++k) ms[k] = a.at(k);
    return ms;
}

}   // namespace

PYBIND11_MODULE(periodic_homogenization, m) {
    m.doc() = "Periodic Homogenization";
    py::module::import("tensors");          // ElasticityTensor{2,3}D, SymmetricMatrix values live there
    py::module detail = m.def_submodule("detail");
    py::class_<HomogenizationResult>(detail, "HomogenizationResult")
        .def_readonly("Ch", &HomogenizationResult::Ch).def_readonly("w_ij", &HomogenizationResult::w_ij).def_readonly("strain_w_ij", &HomogenizationResult::strain_w_ij);
    // homogenize(mesh, Cbase, orthotropicCell = False, manualPeriodicVerticesFile = "", centerFluctuationDisplacements = True,
    //            ignorePeriodicMismatch = False)   (:159-165); device / preconditioner / rtol are this path's additions
#define BIND_HOMOGENIZE(NN)                                                                                                              \
    m.def("homogenize", [](const py::object &mesh, const ETensor<NN> &Cbase, bool orthotropicCell, const std::string &file, bool center, \
                           bool ignoreMismatch, int device, int preconditioner, double rtol) {                                           \
        return homogenizeN<NN>(mesh, Cbase, orthotropicCell, file, center, ignoreMismatch, device, preconditioner, rtol);                 \
    }, py::arg("mesh"), py::arg("Cbase"), py::arg("orthotropicCell") = false, py::arg("manualPeriodicVerticesFile") = std::string(),     \
       py::arg("centerFluctuationDisplacements") = true, py::arg("ignorePeriodicMismatch") = false, py::arg("device") = 0,               \
       py::arg("preconditioner") = (int)MFH_PRECOND_MULTIGRID, py::arg("rtol") = 1e-10);                                                  \
    m.def("probe", [](const py::object &mesh, const ETensor<NN> &Cbase, const py::object &macroStrain, bool orthotropicCell,             \
                      const std::string &file, bool ignoreMismatch) {                                                                   \
        const auto hr = homogenizeN<NN>(mesh, Cbase, orthotropicCell, file, false, ignoreMismatch, 0, (int)MFH_PRECOND_MULTIGRID, 1e-10); \
        return probeN<NN>(mesh, hr, strainOf<NN>(macroStrain));                                                                          \
    }, py::arg("mesh"), py::arg("Cbase"), py::arg("macroStrain"), py::arg("orthotropicCell") = false,                                   \
       py::arg("manualPeriodicVerticesFile") = std::string(), py::arg("ignorePeriodicMismatch") = false)
    BIND_HOMOGENIZE(3);
    BIND_HOMOGENIZE(2);
#undef BIND_HOMOGENIZE
    m.def("probe", [](const py::object &mesh, const HomogenizationResult &hr, const py::object &macroStrain) {
        return hr.w_ij.size() == 6 ? probeN<3>(mesh, hr, strainOf<3>(macroStrain)) : probeN<2>(mesh, hr, strainOf<2>(macroStrain));
    }, py::arg("mesh"), py::arg("homogenizationResult"), py::arg("macroStrain"));
}
