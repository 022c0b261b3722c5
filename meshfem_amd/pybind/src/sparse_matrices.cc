// `sparse_matrices` (== /root/reference/src/python_bindings/sparse_matrices.cc:10-141): Triplet, TripletMatrix, SPSDSystem,
// SuiteSparseMatrix. SPSDSystem / SuiteSparseMatrix.solve hand the matrix to libmeshfem_hip (mfh_matrix_set_upper_triplets): full
// symmetric CSR in HBM, masked Jacobi-preconditioned CG instead of CHOLMOD, the same fixVariables / solve semantics
// (SparseMatrices.hh:2389-2500,2515-2606); constraint rows C x = C_rhs go through a Schur complement around SPD solves.
#include "common.hh"

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <memory>

namespace {

using MeshFEMHip::Triplet;
enum class SMode : uint32_t { NONE = 0, UPPER_TRIANGLE = 1, LOWER_TRIANGLE = 2 };     // SparseMatrices.hh:830

struct TMatrix {
    size_t m = 0, n = 0;
    std::vector<Triplet> nz;
    SMode symmetry_mode = SMode::NONE;
    TMatrix(size_t m_ = 0, size_t n_ = 0) : m(m_), n(n_) {}
    void addNZ(size_t i, size_t j, Real v) {
        if (i >= m || j >= n) throw std::runtime_error("Index out of bounds");
        nz.push_back(Triplet{i, j, v});
    }
    // sort by (column, row), sum duplicates, drop exact zeros (SparseMatrices.hh:280-374)
    void sumRepeated() {
        std::stable_sort(nz.begin(), nz.end(), [](const Triplet &a, const Triplet &b) { return a.j != b.j ? a.j < b.j : a.i < b.i; });
        std::vector<Triplet> out;
        for (const auto &t : nz) {
            if (!out.empty() && out.back().i == t.i && out.back().j == t.j) out.back().v += t.v;
            else out.push_back(t);
        }
        nz.clear();
        for (const auto &t : out) if (t.v != 0.0) nz.push_back(t);
    }
    void reflectUpperTriangle() {
        std::vector<Triplet> out;
        for (const auto &t : nz) if (t.i <= t.j) out.push_back(t);
        const size_t k = out.size();
        for (size_t q = 0; q < k; ++q) if (out[q].i != out[q].j) out.push_back(Triplet{out[q].j, out[q].i, out[q].v});
        nz.swap(out);
        symmetry_mode = SMode::NONE;
    }
    std::vector<Real> diag() const {
        std::vector<Real> d(std::min(m, n), 0.0);
        for (const auto &t : nz) if (t.i == t.j) d[t.i] += t.v;
        return d;
    }
    std::vector<Real> apply(const std::vector<Real> &x) const {
        if (x.size() != n) throw std::runtime_error("Bad vector size");
        std::vector<Real> y(m, 0.0);
        for (const auto &t : nz) {
            y[t.i] += t.v * x[t.j];
            if (symmetry_mode != SMode::NONE && t.i != t.j) y[t.j] += t.v * x[t.i];
        }
        return y;
    }
    // SparseMatrices.hh:623-645: nnz (uint64), row indices, column indices (uint64 each), values (double)
    void dumpBinary(const std::string &path) const {
        std::ofstream os(path, std::ios::binary);
        if (!os) throw std::runtime_error("Failed to open output file " + path);
        const uint64_t N = nz.size();
        os.write((const char *)&N, sizeof(N));
        std::vector<uint64_t> idx(N);
        for (size_t k = 0; k < N; ++k) idx[k] = nz[k].i;
        os.write((const char *)idx.data(), (std::streamsize)(N * sizeof(uint64_t)));
        for (size_t k = 0; k < N; ++k) idx[k] = nz[k].j;
        os.write((const char *)idx.data(), (std::streamsize)(N * sizeof(uint64_t)));
        std::vector<double> v(N);
        for (size_t k = 0; k < N; ++k) v[k] = nz[k].v;
        os.write((const char *)v.data(), (std::streamsize)(N * sizeof(double)));
    }
    void readBinary(const std::string &path) {       // :646-670: the sizes are inferred from the largest indices
        std::ifstream is(path, std::ios::binary);
        if (!is) throw std::runtime_error("Failed to open input file " + path);
        uint64_t N = 0;
        is.read((char *)&N, sizeof(N));
        std::vector<uint64_t> ii(N), jj(N);
        std::vector<double> v(N);
        is.read((char *)ii.data(), (std::streamsize)(N * sizeof(uint64_t)));
        is.read((char *)jj.data(), (std::streamsize)(N * sizeof(uint64_t)));
        is.read((char *)v.data(), (std::streamsize)(N * sizeof(double)));
        if (!is) throw std::runtime_error("Truncated binary matrix file " + path);
        nz.resize(N);
        m = n = 0;
        for (size_t k = 0; k < N; ++k) { nz[k] = Triplet{(size_t)ii[k], (size_t)jj[k], v[k]}; m = std::max(m, (size_t)ii[k] + 1); n = std::max(n, (size_t)jj[k] + 1); }
    }
    void dump(const std::string &path) const {
        std::ofstream os(path);
        os.precision(19);
        for (const auto &t : nz) os << t.i << '\t' << t.j << '\t' << t.v << '\n';
    }
    void rowColRemoval(const std::vector<size_t> &idx) {
        if (m != n) throw std::runtime_error("rowColRemoval is intended for square matrices");
        std::vector<int64_t> renum(m, 0);
        for (size_t i : idx) { if (i >= m) throw std::runtime_error("Index out of bounds"); renum[i] = -1; }
        int64_t k = 0;
        for (auto &r : renum) if (r == 0) r = k++;
        std::vector<Triplet> out;
        for (const auto &t : nz) if (renum[t.i] >= 0 && renum[t.j] >= 0) out.push_back(Triplet{(size_t)renum[t.i], (size_t)renum[t.j], t.v});
        nz.swap(out);
        m = n = (size_t)k;
    }
};

MeshFEMHip::TripletMatrix upperOf(const TMatrix &K) {
    MeshFEMHip::TripletMatrix U;
    U.m = K.m; U.n = K.n;
    for (const auto &t : K.nz)
        if (K.symmetry_mode == SMode::NONE ? t.i <= t.j : true) U.nz.push_back(t.i <= t.j ? t : Triplet{t.j, t.i, t.v});
    return U;
}

// SPSDSystem(K) / SPSDSystem(K, C, C_rhs) (SparseMatrices.hh:2332-2348). With constraint rows the reference factors the KKT
// matrix with UMFPACK; here: x = x0 - K^-1 C^T lambda, (C K^-1 C^T) lambda = C x0 - c, every K^-1 a device PCG solve on the
// free variables (k + 1 solves for k rows).
struct SPSDSys {
    std::unique_ptr<MeshFEMHip::GenericSPSDSystem> sys;
    size_t n = 0;
    std::vector<std::vector<Real>> Crows;   // dense rows of C
    std::vector<Real> Crhs;
    std::vector<size_t> fixedVars;
    std::vector<Real> fixedVals;
    explicit SPSDSys(const TMatrix &K) : sys(new MeshFEMHip::GenericSPSDSystem(upperOf(K))), n(K.m) {}
    SPSDSys(const TMatrix &K, const TMatrix &C, const std::vector<Real> &rhs) : SPSDSys(K) {
        if (C.n != K.n) throw std::runtime_error("Constraint matrix has the wrong number of columns");
        if (rhs.size() != C.m) throw std::runtime_error("Constraint right-hand side has the wrong size");
        Crows.assign(C.m, std::vector<Real>(n, 0.0));
        for (const auto &t : C.nz) Crows[t.i][t.j] += t.v;
        Crhs = rhs;
    }
    void fixVariables(const std::vector<size_t> &vars, const std::vector<Real> &vals, bool /*keepFactorization*/) {
        if (vals.size() != vars.size()) throw std::runtime_error("fixedVars / fixedVarValues size mismatch");
        sys->fixVariables(vars, vals);
        fixedVars.insert(fixedVars.end(), vars.begin(), vars.end());
        fixedVals.insert(fixedVals.end(), vals.begin(), vals.end());
    }
    std::vector<Real> solve(const std::vector<Real> &b) {
        std::vector<Real> x0;
        sys->solve(b, x0);
        const size_t k = Crows.size();
        if (k == 0) return x0;
        // rows restricted to the free variables; the fixed values move to the right-hand side
        std::vector<uint8_t> fixed(n, 0);
        for (size_t v : fixedVars) fixed[v] = 1;
        std::vector<std::vector<Real>> Y(k);                     // K^-1 C_free^T with homogeneous fixed values
        std::vector<Real> zero(n, 0.0);
        for (size_t r = 0; r < k; ++r) {
            std::vector<Real> c = Crows[r];
            for (size_t v = 0; v < n; ++v) if (fixed[v]) c[v] = 0.0;
            Y[r] = solveHomogeneous(c);
        }
        std::vector<Real> S(k * k, 0.0), g(k, 0.0);
        for (size_t r = 0; r < k; ++r) {
            for (size_t q = 0; q < k; ++q) { double s = 0; for (size_t v = 0; v < n; ++v) if (!fixed[v]) s += Crows[r][v] * Y[q][v]; S[r * k + q] = s; }
            double s = 0;
            for (size_t v = 0; v < n; ++v) s += Crows[r][v] * x0[v];
            g[r] = s - Crhs[r];
        }
        // dense solve S lambda = g (k <= a handful)
        for (size_t c = 0; c < k; ++c) {
            size_t p = c;
            for (size_t r = c + 1; r < k; ++r) if (std::fabs(S[r * k + c]) > std::fabs(S[p * k + c])) p = r;
            if (S[p * k + c] == 0.0) throw std::runtime_error("Singular constraint system (redundant constraint rows?)");
            if (p != c) { for (size_t j = 0; j < k; ++j) std::swap(S[p * k + j], S[c * k + j]); std::swap(g[p], g[c]); }
            for (size_t r = c + 1; r < k; ++r) {
                const double f = S[r * k + c] / S[c * k + c];
                for (size_t j = c; j < k; ++j) S[r * k + j] -= f * S[c * k + j];
                g[r] -= f * g[c];
            }
        }
        std::vector<Real> lam(k);
        for (size_t c = k; c-- > 0;) { double s = g[c]; for (size_t j = c + 1; j < k; ++j) s -= S[c * k + j] * lam[j]; lam[c] = s / S[c * k + c]; }
        for (size_t r = 0; r < k; ++r)
            for (size_t v = 0; v < n; ++v) if (!fixed[v]) x0[v] -= lam[r] * Y[r][v];
        return x0;
    }
private:
    std::vector<Real> solveHomogeneous(const std::vector<Real> &rhs) {
        // K^-1 rhs on the free variables with the fixed VALUES taken as 0: solve with the actual values and subtract the
        // solution of the zero right-hand side (both linear in the data)
        std::vector<Real> a, z;
        sys->solve(rhs, a);
        if (fixedVars.empty()) return a;
        sys->solve(std::vector<Real>(n, 0.0), z);
        for (size_t v = 0; v < n; ++v) a[v] -= z[v];
        return a;
    }
};

// Compressed-column matrix (SparseMatrices.hh:1380-1500), 64-bit indices like SuiteSparse_long
struct SSMatrix {
    int64_t m = 0, n = 0, nz = 0;
    std::vector<int64_t> Ap{0}, Ai;
    std::vector<Real> Ax;
    SMode symmetry_mode = SMode::NONE;
    SSMatrix() = default;
    explicit SSMatrix(const TMatrix &T) { setFromTMatrix(T); }
    explicit SSMatrix(const std::string &path) { readBinary(path); }
    void setFromTMatrix(const TMatrix &Tin) {
        TMatrix T = Tin;
        T.sumRepeated();
        m = (int64_t)T.m; n = (int64_t)T.n; nz = (int64_t)T.nz.size();
        symmetry_mode = T.symmetry_mode;
        Ap.assign((size_t)n + 1, 0); Ai.resize((size_t)nz); Ax.resize((size_t)nz);
        for (const auto &t : T.nz) ++Ap[t.j + 1];
        for (int64_t j = 0; j < n; ++j) Ap[(size_t)j + 1] += Ap[(size_t)j];
        for (size_t k = 0; k < T.nz.size(); ++k) { Ai[k] = (int64_t)T.nz[k].i; Ax[k] = T.nz[k].v; }   // already in (col, row) order
    }
    TMatrix getTripletMatrix() const {
        TMatrix T((size_t)m, (size_t)n);
        T.symmetry_mode = symmetry_mode;
        for (int64_t j = 0; j < n; ++j)
            for (int64_t k = Ap[(size_t)j]; k < Ap[(size_t)j + 1]; ++k) T.nz.push_back(Triplet{(size_t)Ai[(size_t)k], (size_t)j, Ax[(size_t)k]});
        return T;
    }
    Real trace() const {
        Real t = 0;
        for (int64_t j = 0; j < n; ++j)
            for (int64_t k = Ap[(size_t)j]; k < Ap[(size_t)j + 1]; ++k) if (Ai[(size_t)k] == j) t += Ax[(size_t)k];
        return t;
    }
    std::vector<Real> apply(const std::vector<Real> &x, bool transpose) const {
        if ((int64_t)x.size() != (transpose ? m : n)) throw std::runtime_error("Bad vector size");
        std::vector<Real> y((size_t)(transpose ? n : m), 0.0);
        for (int64_t j = 0; j < n; ++j)
            for (int64_t k = Ap[(size_t)j]; k < Ap[(size_t)j + 1]; ++k) {
                const int64_t i = Ai[(size_t)k];
                if (transpose) y[(size_t)j] += Ax[(size_t)k] * x[(size_t)i]; else y[(size_t)i] += Ax[(size_t)k] * x[(size_t)j];
                if (symmetry_mode != SMode::NONE && i != j) { if (transpose) y[(size_t)i] += Ax[(size_t)k] * x[(size_t)j]; else y[(size_t)j] += Ax[(size_t)k] * x[(size_t)i]; }
            }
        return y;
    }
    // SparseMatrices.hh:1448-1495: m, n, nz (long), symmetry mode (int), Ap, Ai, Ax
    void dumpBinary(const std::string &path) const {
        std::ofstream os(path, std::ios::binary);
        if (!os) throw std::runtime_error("Couldn't open output file " + path);
        const int64_t hdr[3] = {m, n, nz};
        const uint32_t sm = (uint32_t)symmetry_mode;
        os.write((const char *)hdr, sizeof(hdr)); os.write((const char *)&sm, sizeof(sm));
        os.write((const char *)Ap.data(), (std::streamsize)(Ap.size() * sizeof(int64_t)));
        os.write((const char *)Ai.data(), (std::streamsize)(Ai.size() * sizeof(int64_t)));
        os.write((const char *)Ax.data(), (std::streamsize)(Ax.size() * sizeof(Real)));
    }
    void readBinary(const std::string &path) {
        std::ifstream is(path, std::ios::binary);
        if (!is) throw std::runtime_error("Couldn't open input file " + path);
        int64_t hdr[3]; uint32_t sm;
        is.read((char *)hdr, sizeof(hdr)); is.read((char *)&sm, sizeof(sm));
        if (sm > 2) throw std::runtime_error("Invalid symmetry_mode");
        m = hdr[0]; n = hdr[1]; nz = hdr[2]; symmetry_mode = (SMode)sm;
        Ap.resize((size_t)n + 1); Ai.resize((size_t)nz); Ax.resize((size_t)nz);
        is.read((char *)Ap.data(), (std::streamsize)(Ap.size() * sizeof(int64_t)));
        is.read((char *)Ai.data(), (std::streamsize)(Ai.size() * sizeof(int64_t)));
        is.read((char *)Ax.data(), (std::streamsize)(Ax.size() * sizeof(Real)));
        if (!is) throw std::runtime_error("Truncated binary matrix file " + path);
    }
};

ArrD toArr(const std::vector<Real> &v) { ArrD o((py::ssize_t)v.size()); std::copy(v.begin(), v.end(), o.mutable_data()); return o; }
std::vector<Real> toVec(const ArrD &a) { return std::vector<Real>(a.data(), a.data() + a.size()); }

}   // namespace

PYBIND11_MODULE(sparse_matrices, m) {
    m.doc() = "Sparse Representations and Solvers";
    py::class_<Triplet>(m, "Triplet")
        .def(py::init([](size_t i, size_t j, Real v) { return Triplet{i, j, v}; }))
        .def_readwrite("i", &Triplet::i).def_readwrite("j", &Triplet::j).def_readwrite("v", &Triplet::v);
    py::enum_<SMode>(m, "SymmetryMode").value("NONE", SMode::NONE).value("UPPER_TRIANGLE", SMode::UPPER_TRIANGLE).value("LOWER_TRIANGLE", SMode::LOWER_TRIANGLE);

    py::class_<TMatrix>(m, "TripletMatrix", "Sparse matrix in triplet (COO) format")
        .def(py::init<size_t, size_t>(), py::arg("m") = 0, py::arg("n") = 0)
        .def_property_readonly("nnz", [](const TMatrix &A) { return A.nz.size(); })
        .def_property_readonly("m", [](const TMatrix &A) { return A.m; })
        .def_property_readonly("n", [](const TMatrix &A) { return A.n; })
        .def_readwrite("symmetry_mode", &TMatrix::symmetry_mode)
        .def("entries", [](const TMatrix &A) { return py::make_iterator(A.nz.cbegin(), A.nz.cend()); }, py::keep_alive<0, 1>())
        .def("addNZ", &TMatrix::addNZ, "Add a triplet to the matrix")
        .def("reflectUpperTriangle", &TMatrix::reflectUpperTriangle)
        .def("diag", [](const TMatrix &A) { return toArr(A.diag()); })
        .def("rowColRemoval", &TMatrix::rowColRemoval)
        .def("sumRepeated", &TMatrix::sumRepeated)
        .def("apply", [](const TMatrix &A, const ArrD &x) { return toArr(A.apply(toVec(x))); })
        .def("arrays", [](const TMatrix &A) {          // (i, j, v) as numpy arrays (what compressedColumn / toSciPy callers build on)
            ArrI i((py::ssize_t)A.nz.size()), j((py::ssize_t)A.nz.size());
            ArrD v((py::ssize_t)A.nz.size());
            for (size_t k = 0; k < A.nz.size(); ++k) { i.mutable_at(k) = (int64_t)A.nz[k].i; j.mutable_at(k) = (int64_t)A.nz[k].j; v.mutable_at(k) = A.nz[k].v; }
            return py::make_tuple(i, j, v);
        })
        .def("compressedColumn", [](const TMatrix &A) {
            py::object csc = py::module::import("scipy.sparse").attr("csc_matrix");
            ArrI i((py::ssize_t)A.nz.size()), j((py::ssize_t)A.nz.size());
            ArrD v((py::ssize_t)A.nz.size());
            for (size_t k = 0; k < A.nz.size(); ++k) { i.mutable_at(k) = (int64_t)A.nz[k].i; j.mutable_at(k) = (int64_t)A.nz[k].j; v.mutable_at(k) = A.nz[k].v; }
            return csc(py::make_tuple(v, py::make_tuple(i, j)), py::make_tuple(A.m, A.n));
        })
        .def("dump", &TMatrix::dump).def("dumpBinary", &TMatrix::dumpBinary).def("readBinary", &TMatrix::readBinary);

    py::class_<SPSDSys>(m, "SPSDSystem", "A (constrained) SPSD system that can be solved for several different right-hand sides.")
        .def(py::init<const TMatrix &>(), py::arg("K"))
        .def(py::init<const TMatrix &, const TMatrix &, const std::vector<Real> &>(), py::arg("K"), py::arg("C"), py::arg("C_rhs"))
        .def("fixVariables", &SPSDSys::fixVariables, py::arg("fixedVars"), py::arg("fixedVarValues"), py::arg("keepFactorization") = false)
        .def("setForceSupernodal", [](SPSDSys &, bool) {}, "No-op: there is no factorisation (PCG on the device)")
        .def("solve", [](SPSDSys &s, const ArrD &b) { return toArr(s.solve(toVec(b))); })
        .def_property("rtol", [](const SPSDSys &s) { return s.sys->rtol; }, [](SPSDSys &s, double r) { s.sys->rtol = r; })
        .def_property_readonly("iterations", [](const SPSDSys &s) { return s.sys->info.iterations; });

    py::class_<SSMatrix, std::shared_ptr<SSMatrix>>(m, "SuiteSparseMatrix", "Sparse matrix in a Suite Sparse-compatible compressed column format")
        .def(py::init<const TMatrix &>(), py::arg("tripletMatrix"))
        .def(py::init<const std::string &>(), py::arg("bin_dump_path"))
        .def(py::init<>())
        .def_readwrite("m", &SSMatrix::m).def_readwrite("n", &SSMatrix::n).def_readwrite("nz", &SSMatrix::nz)
        .def_readwrite("Ap", &SSMatrix::Ap).def_readwrite("Ai", &SSMatrix::Ai).def_readwrite("Ax", &SSMatrix::Ax)
        .def_readwrite("symmetry_mode", &SSMatrix::symmetry_mode)
        .def("trace", &SSMatrix::trace)
        .def("setFromTMatrix", &SSMatrix::setFromTMatrix)
        .def("getTripletMatrix", &SSMatrix::getTripletMatrix)
        .def("apply", [](const SSMatrix &A, const ArrD &x, bool transpose) { return toArr(A.apply(toVec(x), transpose)); }, py::arg("vec"), py::arg("transpose") = false)
        .def("toSciPy", [](const SSMatrix &A) {
            py::object csc = py::module::import("scipy.sparse").attr("csc_matrix");
            return csc(py::make_tuple(toArr(A.Ax), ArrI(py::cast(A.Ai)), ArrI(py::cast(A.Ap))), py::make_tuple(A.m, A.n));
        })
        .def("solve", [](const SSMatrix &A, const ArrD &b) {
            if (A.symmetry_mode != SMode::UPPER_TRIANGLE) throw std::runtime_error("Only symmetric matrices are currently supported");
            SPSDSys s(A.getTripletMatrix());
            return toArr(s.solve(toVec(b)));
        })
        .def("dumpBinary", &SSMatrix::dumpBinary).def("readBinary", &SSMatrix::readBinary)
        .def(py::pickle([](const SSMatrix &A) { return py::make_tuple(A.m, A.n, A.nz, A.Ap, A.Ai, A.Ax, (int)A.symmetry_mode); },
                        [](const py::tuple &t) {
                            if (t.size() != 7) throw std::runtime_error("Invalid state!");
                            SSMatrix A;
                            A.m = t[0].cast<int64_t>(); A.n = t[1].cast<int64_t>(); A.nz = t[2].cast<int64_t>();
                            A.Ap = t[3].cast<std::vector<int64_t>>(); A.Ai = t[4].cast<std::vector<int64_t>>(); A.Ax = t[5].cast<std::vector<Real>>();
                            A.symmetry_mode = (SMode)t[6].cast<int>();
                            return A;
                        }));
}
