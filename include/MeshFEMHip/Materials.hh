////////////////////////////////////////////////////////////////////////////////
// MeshFEMHip/Materials.hh
////////////////////////////////////////////////////////////////////////////////
// `.material` files for the C++ facade: Materials::Constant<N> with the reference's interface (Materials.hh:59-103,
// Materials.cc:183-328) over a small ElasticityTensor<N> holding the flattened tensor D (Voigt order xx,yy,zz,yz,xz,xy,
// TENSOR shear entries: ElasticityTensor.hh:100-164, Flattening.hh:47-60). Types "isotropic[_material]",
// "orthotropic[_material]" and "symmetric_material" / "anisotropic"; the same symmetry checks and messages.
//
//     Materials::Constant<3> mat("B9Creator.material");     // or mat.setFromFile(path) / setFromJson(json)
//     sim.setMaterial(mat);                                  // -> mfh_material_const
#ifndef MESHFEMHIP_MATERIALS_HH
#define MESHFEMHIP_MATERIALS_HH

#include <array>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Json.hh"

namespace MeshFEMHip {

constexpr size_t flatLen(size_t N) { return N * (N + 1) / 2; }

template <size_t N>
class ElasticityTensor {
public:
    static constexpr size_t FL = flatLen(N);
    ElasticityTensor(double E = 1.0, double nu = 0.0) { setIsotropic(E, nu); }

    double &D(size_t i, size_t j) { return i <= j ? m_D[i * FL + j] : m_D[j * FL + i]; }    // upper triangle is the storage
    double D(size_t i, size_t j) const { return i <= j ? m_D[i * FL + j] : m_D[j * FL + i]; }

    // ElasticityTensor.hh:100-134; 2D = plane stress
    void setIsotropic(double E, double nu) {
        double lambda = N == 2 ? (nu * E) / (1.0 - nu * nu) : (nu * E) / ((1.0 + nu) * (1.0 - 2.0 * nu));
        setIsotropicLame(lambda, E / (2.0 + 2.0 * nu));
    }
    void setIsotropicLame(double lambda, double mu) {
        m_D.fill(0.0);
        for (size_t i = 0; i < N; ++i)
            for (size_t j = i; j < N; ++j) D(i, j) = lambda + (i == j ? 2.0 * mu : 0.0);
        for (size_t i = N; i < FL; ++i) D(i, i) = mu;
    }
    // ElasticityTensor.hh:136-164: inverse of the orthotropic compliance
    void setOrthotropic3D(double Ex, double Ey, double Ez, double nuYX, double nuZX, double nuZY, double muYZ, double muZX, double muXY) {
        static_assert(N == 3 || N == 2, "dimension");
        if (N != 3) throw std::runtime_error("setOrthotropic3D on a 2D tensor");
        std::array<double, 36> S{};
        auto s = [&](size_t i, size_t j) -> double & { return S[i * 6 + j]; };
        s(0, 0) = 1.0 / Ex; s(0, 1) = s(1, 0) = -nuYX / Ey; s(0, 2) = s(2, 0) = -nuZX / Ez;
        s(1, 1) = 1.0 / Ey; s(1, 2) = s(2, 1) = -nuZY / Ez; s(2, 2) = 1.0 / Ez;
        s(3, 3) = 1.0 / muYZ; s(4, 4) = 1.0 / muZX; s(5, 5) = 1.0 / muXY;
        m_setInverse(S.data(), 6);
    }
    void setOrthotropic2D(double Ex, double Ey, double nuYX, double muXY) {
        if (N != 2) throw std::runtime_error("setOrthotropic2D on a 3D tensor");
        std::array<double, 9> S{};
        S[0] = 1.0 / Ex; S[1] = S[3] = -nuYX / Ey; S[4] = 1.0 / Ey; S[8] = 1.0 / muXY;
        m_setInverse(S.data(), 3);
    }
    // full flatLen x flatLen matrix, row-major: what mfh_material_const takes
    std::vector<double> flat() const {
        std::vector<double> out(FL * FL);
        for (size_t i = 0; i < FL; ++i) for (size_t j = 0; j < FL; ++j) out[i * FL + j] = D(i, j);
        return out;
    }

private:
    // Gauss-Jordan with partial pivoting on the (symmetric positive definite) compliance
    void m_setInverse(const double *S, size_t n) {
        std::vector<double> a(S, S + n * n), inv(n * n, 0.0);
        for (size_t i = 0; i < n; ++i) inv[i * n + i] = 1.0;
        for (size_t c = 0; c < n; ++c) {
            size_t p = c;
            for (size_t r = c + 1; r < n; ++r) if (std::fabs(a[r * n + c]) > std::fabs(a[p * n + c])) p = r;
            if (a[p * n + c] == 0.0) throw std::runtime_error("Singular compliance tensor");
            if (p != c)
                for (size_t k = 0; k < n; ++k) { std::swap(a[p * n + k], a[c * n + k]); std::swap(inv[p * n + k], inv[c * n + k]); }
            double d = 1.0 / a[c * n + c];
            for (size_t k = 0; k < n; ++k) { a[c * n + k] *= d; inv[c * n + k] *= d; }
            for (size_t r = 0; r < n; ++r) {
                if (r == c) continue;
                double f = a[r * n + c];
                if (f == 0.0) continue;
                for (size_t k = 0; k < n; ++k) { a[r * n + k] -= f * a[c * n + k]; inv[r * n + k] -= f * inv[c * n + k]; }
            }
        }
        for (size_t i = 0; i < n; ++i) for (size_t j = i; j < n; ++j) D(i, j) = 0.5 * (inv[i * n + j] + inv[j * n + i]);
    }
    std::array<double, FL * FL> m_D{};
};

namespace Materials {

template <size_t N>
class Constant {
public:
    using ETensor = ElasticityTensor<N>;
    Constant() {}
    explicit Constant(const std::string &materialPath) { setFromFile(materialPath); }
    explicit Constant(const ETensor &E) : m_E(E) {}

    void setTensor(const ETensor &E) { m_E = E; }
    const ETensor &getTensor() const { return m_E; }
    void setIsotropic(double E, double nu) { m_E.setIsotropic(E, nu); }
    void setOrthotropic2D(double Ex, double Ey, double nuYX, double muXY) { m_E.setOrthotropic2D(Ex, Ey, nuYX, muXY); }
    void setOrthotropic3D(double Ex, double Ey, double Ez, double nuYX, double nuZX, double nuZY, double muYZ, double muZX, double muXY) {
        m_E.setOrthotropic3D(Ex, Ey, Ez, nuYX, nuZX, nuZY, muYZ, muZX, muXY);
    }

    void setFromFile(const std::string &materialPath) {            // Materials.cc:301-311
        std::ifstream is(materialPath);
        if (!is.is_open()) throw std::runtime_error("Couldn't open material " + materialPath);
        is.close();
        setFromJson(Json::parseFile(materialPath));
    }
    void setFromJson(const Json &config) {                           // :287-298
        const std::string type = config["type"].string();
        if (type == "isotropic_material" || type == "isotropic") m_parseIsotropic(config);
        else if (type == "orthotropic_material" || type == "orthotropic") m_parseOrthotropic(config);
        else if (type == "symmetric_material" || type == "anisotropic") m_parseAnisotropic(config);
        else throw std::runtime_error("Invalid type.");
    }
    // always the anisotropic form (:313-328)
    std::string getJsonString() const {
        std::string s = "{\"material_matrix\":[";
        char buf[40];
        for (size_t i = 0; i < flatLen(N); ++i) {
            s += i ? ",[" : "[";
            for (size_t j = 0; j < flatLen(N); ++j) {
                snprintf(buf, sizeof(buf), "%s%.17g", j ? "," : "", m_E.D(i, j));
                s += buf;
            }
            s += "]";
        }
        return s + "],\"type\":\"anisotropic\"}";
    }

private:
    static std::vector<double> m_parseNVector(size_t n, const Json &entry) {        // :183-188
        std::vector<double> v;
        for (const auto &x : entry.items()) v.push_back(x.number());
        if (!entry.is_array() || v.size() != n) throw std::runtime_error("Failed to parse vector of size " + std::to_string(n));
        return v;
    }
    void m_parseIsotropic(const Json &entry) { m_E.setIsotropic(entry["young"].number(), entry["poisson"].number()); }   // :193-198
    void m_parseOrthotropic(const Json &entry) {                     // :210-245
        if (N == 2) {
            auto young = m_parseNVector(2, entry["young"]), poisson = m_parseNVector(2, entry["poisson"]), shear = m_parseNVector(1, entry["shear"]);
            double Ex = young[0], Ey = young[1], nuXY = poisson[0], nuYX = poisson[1];
            m_E.setOrthotropic2D(Ex, Ey, nuYX, shear[0]);
            if (std::fabs(nuYX / Ey - nuXY / Ex) > 1e-10) throw std::runtime_error("Orthotopic parameters violate symmetry");
        } else {
            auto young = m_parseNVector(3, entry["young"]), poisson = m_parseNVector(6, entry["poisson"]), shear = m_parseNVector(3, entry["shear"]);
            double Ex = young[0], Ey = young[1], Ez = young[2];
            double nuYZ = poisson[0], nuZY = poisson[1], nuZX = poisson[2], nuXZ = poisson[3], nuXY = poisson[4], nuYX = poisson[5];
            m_E.setOrthotropic3D(Ex, Ey, Ez, nuYX, nuZX, nuZY, shear[0], shear[1], shear[2]);
            if (std::fabs(nuYX / Ey - nuXY / Ex) > 1e-10 || std::fabs(nuYZ / Ey - nuZY / Ez) > 1e-10 || std::fabs(nuZX / Ez - nuXZ / Ex) > 1e-10)
                throw std::runtime_error("Orthotopic parameters violate symmetry");
        }
    }
    void m_parseAnisotropic(const Json &entry) {                     // :256-273: upper triangle kept, lower checked against it
        size_t row = 0;
        for (const auto &r : entry["material_matrix"].items()) {
            if (r.size() != flatLen(N) || row >= flatLen(N)) throw std::runtime_error("Failed to parse material_matrix");
            for (size_t col = 0; col < flatLen(N); ++col) {
                double val = r[col].number();
                if (row <= col) m_E.D(row, col) = val;
                else if (std::fabs(m_E.D(row, col) - val) > 1e-10) throw std::runtime_error("Asymmetric material_matrix");
            }
            ++row;
        }
    }
    ETensor m_E;
};

} // namespace Materials
} // namespace MeshFEMHip

#endif /* end of include guard: MESHFEMHIP_MATERIALS_HH */
