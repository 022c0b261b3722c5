// `tensors` (== /root/reference/src/python_bindings/tensors.cc:13-127): ElasticityTensor2D / ElasticityTensor3D,
// SymmetricMatrix values. Host-side value types; the device consumes their flattened D through mfh_material_const.
#include "common.hh"

namespace {

// dense inverse of a small SPD / regular matrix (Gauss-Jordan with partial pivoting)
template <size_t M> bool invertSmall(double (&A)[M][M]) {
    double B[M][2 * M];
    for (size_t i = 0; i < M; ++i)
        for (size_t j = 0; j < M; ++j) { B[i][j] = A[i][j]; B[i][M + j] = i == j ? 1.0 : 0.0; }
    for (size_t c = 0; c < M; ++c) {
        size_t p = c;
        for (size_t r = c + 1; r < M; ++r) if (std::fabs(B[r][c]) > std::fabs(B[p][c])) p = r;
        if (B[p][c] == 0.0) return false;
        if (p != c) for (size_t j = 0; j < 2 * M; ++j) std::swap(B[p][j], B[c][j]);
        const double inv = 1.0 / B[c][c];
        for (size_t j = 0; j < 2 * M; ++j) B[c][j] *= inv;
        for (size_t r = 0; r < M; ++r) {
            if (r == c) continue;
            const double f = B[r][c];
            if (f != 0.0) for (size_t j = 0; j < 2 * M; ++j) B[r][j] -= f * B[c][j];
        }
    }
    for (size_t i = 0; i < M; ++i) for (size_t j = 0; j < M; ++j) A[i][j] = B[i][M + j];
    return true;
}

template <size_t N> ArrD tensorD(const ETensor<N> &E) {
    ArrD out = make2d(ETensor<N>::FL, ETensor<N>::FL);
    auto o = out.template mutable_unchecked<2>();
    for (size_t i = 0; i < ETensor<N>::FL; ++i) for (size_t j = 0; j < ETensor<N>::FL; ++j) o(i, j) = E.D[i][j];
    return out;
}

template <size_t N> void bindTensors(py::module &m, py::module &detail) {
    using ET = ETensor<N>;
    using SM = SMValue<N>;
    constexpr size_t FL = ET::FL;
    const std::string dimName = std::to_string(N) + "D";
    auto cls = py::class_<ET>(m, ("ElasticityTensor" + dimName).c_str())
        .def(py::init<>())
        .def(py::init<Real, Real>(), py::arg("E"), py::arg("nu"))
        .def("setIsotropic", &ET::setIsotropic, py::arg("E"), py::arg("nu"))
        .def("setIdentity", [](ET &E) -> ET & { E.setIdentity(); return E; }, py::return_value_policy::reference_internal)
        .def("__call__", [](const ET &E, size_t i, size_t j, size_t k, size_t l) {
            if (i >= N || j >= N || k >= N || l >= N) throw std::runtime_error("Index out of bounds");
            return E(i, j, k, l);
        })
        .def_property("D", [](const ET &E) { return tensorD(E); },
                      [](ET &E, const ArrD &D) {      // (the reference's D is read-only; the setter serves the homogenization result)
                          if (D.ndim() != 2 || (size_t)D.shape(0) != FL || (size_t)D.shape(1) != FL) throw std::runtime_error("D must be flatLen x flatLen");
                          auto d = D.unchecked<2>();
                          for (size_t i = 0; i < FL; ++i) for (size_t j = 0; j < FL; ++j) E.D[i][j] = d(i, j);
                      })
        .def("doubleContract", [](const ET &E, const SM &s) { SM out; out.flat = E.doubleContract(s.flat); return out; }, py::arg("smat"))
        .def("doubleContract", [](const ET &E, const ArrD &e) {
            if (e.ndim() == 1 && (size_t)e.shape(0) == FL) {
                std::array<Real, FL> in;
                for (size_t k = 0; k < FL; ++k) in[k] = e.at(k);
                const auto r = E.doubleContract(in);
                ArrD out((py::ssize_t)FL);
                for (size_t k = 0; k < FL; ++k) out.mutable_at(k) = r[k];
                return out;
            }
            if (e.ndim() != 2 || (size_t)e.shape(1) != FL) throw std::runtime_error("expected a flattened symmetric matrix (field)");
            ArrD out = make2d((size_t)e.shape(0), FL);       // SymmetricMatrixField: one contraction per row
            auto in = e.unchecked<2>();
            auto o = out.mutable_unchecked<2>();
            for (py::ssize_t r = 0; r < e.shape(0); ++r) {
                std::array<Real, FL> v;
                for (size_t k = 0; k < FL; ++k) v[k] = in(r, k);
                const auto w = E.doubleContract(v);
                for (size_t k = 0; k < FL; ++k) o(r, k) = w[k];
            }
            return out;
        }, py::arg("smat"))
        .def("inverse", [](const ET &E) {
            // compliance in the same flattening: S = W^-1 D^-1 W^-1, W = shear doubling (ElasticityTensor.hh:437-449)
            double A[FL][FL];
            for (size_t i = 0; i < FL; ++i) for (size_t j = 0; j < FL; ++j) A[i][j] = E.D[i][j] * (j < N ? 1.0 : 2.0);
            if (!invertSmall<FL>(A)) throw std::runtime_error("singular elasticity tensor");
            ET out;
            for (size_t i = 0; i < FL; ++i) for (size_t j = 0; j < FL; ++j) out.D[i][j] = A[i][j] / (j < N ? 1.0 : 2.0);
            return out;
        })
        .def("frobeniusNormSq", [](const ET &E) {
            double s = 0;
            for (size_t i = 0; i < FL; ++i) for (size_t j = 0; j < FL; ++j) s += E.D[i][j] * E.D[i][j] * (i < N ? 1.0 : 2.0) * (j < N ? 1.0 : 2.0);
            return s;
        })
        .def("__sub__", [](const ET &a, const ET &b) { ET o; for (size_t i = 0; i < FL; ++i) for (size_t j = 0; j < FL; ++j) o.D[i][j] = a.D[i][j] - b.D[i][j]; return o; })
        .def("__repr__", [](const ET &E) {
            std::stringstream ss;
            ss << N << "D elasticity tensor, flattened D =\n";
            for (size_t i = 0; i < FL; ++i) { for (size_t j = 0; j < FL; ++j) ss << (j ? "\t" : "") << E.D[i][j]; ss << "\n"; }
            return ss.str();
        });
    if (N == 3)
        cls.def("setOrthotropic", [](ET &E, Real Ex, Real Ey, Real Ez, Real nuYX, Real nuZX, Real nuZY, Real muYZ, Real muZX, Real muXY) {
            // ElasticityTensor.hh:136-152: compliance-like matrix, symmetrised, inverted
            double Sm[6][6] = {};
            Sm[0][0] = 1 / Ex; Sm[0][1] = -nuYX / Ey; Sm[0][2] = -nuZX / Ez; Sm[1][1] = 1 / Ey; Sm[1][2] = -nuZY / Ez; Sm[2][2] = 1 / Ez;
            Sm[3][3] = 1 / muYZ; Sm[4][4] = 1 / muZX; Sm[5][5] = 1 / muXY;
            for (int i = 0; i < 6; ++i) for (int j = 0; j < i; ++j) Sm[i][j] = Sm[j][i];
            if (!invertSmall<6>(Sm)) throw std::runtime_error("singular orthotropic parameters");
            for (size_t i = 0; i < FL; ++i) for (size_t j = 0; j < FL; ++j) E.D[i][j] = Sm[i][j];
        }, py::arg("Ex"), py::arg("Ey"), py::arg("Ez"), py::arg("nuYX"), py::arg("nuZX"), py::arg("nuZY"), py::arg("muYZ"), py::arg("myZX"), py::arg("muXY"));
    else
        cls.def("setOrthotropic", [](ET &E, Real Ex, Real Ey, Real nuYX, Real muXY) {           // :154-164
            double Sm[3][3] = {};
            Sm[0][0] = 1 / Ex; Sm[0][1] = Sm[1][0] = -nuYX / Ey; Sm[1][1] = 1 / Ey; Sm[2][2] = 1 / muXY;
            if (!invertSmall<3>(Sm)) throw std::runtime_error("singular orthotropic parameters");
            for (size_t i = 0; i < FL; ++i) for (size_t j = 0; j < FL; ++j) E.D[i][j] = Sm[i][j];
        }, py::arg("Ex"), py::arg("Ey"), py::arg("nuYX"), py::arg("muXY"));

    py::class_<SM>(detail, ("SymmetricMatrixValue" + dimName).c_str())
        .def("__call__", [](const SM &s, size_t i, size_t j) { if (i >= N || j >= N) throw std::runtime_error("Index out of bounds"); return s(i, j); }, py::arg("i"), py::arg("j"))
        .def("__getitem__", [](const SM &s, size_t k) { if (k >= FL) throw std::out_of_range("flattened index"); return s.flat[k]; })
        .def_property_readonly("flat", [](const SM &s) { ArrD o((py::ssize_t)FL); for (size_t k = 0; k < FL; ++k) o.mutable_at(k) = s.flat[k]; return o; })
        .def_property_readonly("N", [](const SM &) { return N; })
        .def("toMatrix", [](const SM &s) {
            ArrD o = make2d(N, N);
            auto v = o.mutable_unchecked<2>();
            for (size_t i = 0; i < N; ++i) for (size_t j = 0; j < N; ++j) v(i, j) = s(i, j);
            return o;
        });
}

}   // namespace

PYBIND11_MODULE(tensors, m) {
    m.doc() = "Tensors and tensor fields used for elasticity simulations";
    py::module detail = m.def_submodule("detail");
    bindTensors<2>(m, detail);
    bindTensors<3>(m, detail);
    // SymmetricMatrix(flatValues) / SymmetricMatrix(mat) (tensors.cc:108-109)
    m.def("SymmetricMatrix", [](const ArrD &a) -> py::object {
        if (a.ndim() == 2) {
            const size_t d = (size_t)a.shape(0);
            if ((d != 2 && d != 3) || (size_t)a.shape(1) != d) throw std::runtime_error("expected a 2x2 or 3x3 matrix");
            auto v = a.unchecked<2>();
            if (d == 3) { SMValue<3> s; for (size_t i = 0; i < 3; ++i) for (size_t j = i; j < 3; ++j) s.flat[flattenIndices<3>(i, j)] = v(i, j); return py::cast(s); }
            SMValue<2> s; for (size_t i = 0; i < 2; ++i) for (size_t j = i; j < 2; ++j) s.flat[flattenIndices<2>(i, j)] = v(i, j); return py::cast(s);
        }
        if (a.ndim() != 1 || (a.shape(0) != 3 && a.shape(0) != 6)) throw std::runtime_error("expected 3 or 6 flattened values");
        if (a.shape(0) == 6) { SMValue<3> s; for (size_t k = 0; k < 6; ++k) s.flat[k] = a.at(k); return py::cast(s); }
        SMValue<2> s; for (size_t k = 0; k < 3; ++k) s.flat[k] = a.at(k); return py::cast(s);
    }, py::arg("values"));
}
