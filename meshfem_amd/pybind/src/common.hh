// Shared helpers of the compiled pybind11 modules (mesh, tensors, sparse_matrices, periodic_homogenization): the names, argument
// lists and behaviour of the reference's extension modules (/root/reference/src/python_bindings/*.cc) over the C ABI of
// libmeshfem_hip (include/meshfem_hip.h) and the C++ facade (include/MeshFEMHip/*.hh). numpy arrays replace Eigen matrices
// (Eigen is not installed in this image; pybind11/eigen.h would be the one-line change with it).
#pragma once
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <array>
#include <cmath>
#include <cstdint>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/MeshFEMHip/LinearElasticity.hh"
#include "../../../include/MeshFEMHip/PeriodicHomogenization.hh"

namespace py = pybind11;
using Real = double;
using ArrD = py::array_t<double, py::array::c_style | py::array::forcecast>;
using ArrI = py::array_t<int64_t, py::array::c_style | py::array::forcecast>;

constexpr size_t flatLenOf(size_t N) { return N * (N + 1) / 2; }
// Flattening.hh:47-60: 3D xx,yy,zz,yz,xz,xy ; 2D xx,yy,xy
template <size_t N> inline size_t flattenIndices(size_t i, size_t j) { return i == j ? i : (N * (N + 1) / 2 - i - j); }

// ---- tensors shared between the `tensors` module (which owns the Python classes) and its users
template <size_t N> struct ETensor {
    static constexpr size_t FL = flatLenOf(N);
    double D[FL][FL];
    ETensor() { for (auto &r : D) for (double &v : r) v = 0.0; }
    ETensor(Real E, Real nu) { setIsotropic(E, nu); }
    void setIsotropic(Real E, Real nu) {                  // ElasticityTensor.hh:100-134 (2D: plane stress)
        for (auto &r : D) for (double &v : r) v = 0.0;
        const Real lam = N == 3 ? (nu * E) / ((1.0 + nu) * (1.0 - 2.0 * nu)) : (nu * E) / (1.0 - nu * nu), mu = E / (2.0 + 2.0 * nu);
        for (size_t i = 0; i < N; ++i) {
            for (size_t j = 0; j < N; ++j) D[i][j] = lam;
            D[i][i] = lam + 2 * mu;
        }
        for (size_t k = N; k < FL; ++k) D[k][k] = mu;
    }
    void setIdentity() {
        for (auto &r : D) for (double &v : r) v = 0.0;
        for (size_t i = 0; i < N; ++i) D[i][i] = 1.0;
        for (size_t k = N; k < FL; ++k) D[k][k] = 0.5;
    }
    Real operator()(size_t i, size_t j, size_t k, size_t l) const { return D[flattenIndices<N>(i, j)][flattenIndices<N>(k, l)]; }
    std::vector<Real> flat() const {
        std::vector<Real> out(FL * FL);
        for (size_t i = 0; i < FL; ++i) for (size_t j = 0; j < FL; ++j) out[i * FL + j] = D[i][j];
        return out;
    }
    std::array<Real, FL> doubleContract(const std::array<Real, FL> &e) const {   // :444-449: D * shearDoubled(e)
        std::array<Real, FL> out{};
        for (size_t i = 0; i < FL; ++i)
            for (size_t j = 0; j < FL; ++j) out[i] += D[i][j] * (j < N ? 1.0 : 2.0) * e[j];
        return out;
    }
};

template <size_t N> struct SMValue {
    static constexpr size_t FL = flatLenOf(N);
    std::array<Real, FL> flat{};
    Real operator()(size_t i, size_t j) const { return flat[flattenIndices<N>(i, j)]; }
};

inline ArrD make2d(size_t r, size_t c) { return ArrD(std::vector<py::ssize_t>{(py::ssize_t)r, (py::ssize_t)c}); }
