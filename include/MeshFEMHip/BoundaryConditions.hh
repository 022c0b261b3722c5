////////////////////////////////////////////////////////////////////////////////
// MeshFEMHip/BoundaryConditions.hh
////////////////////////////////////////////////////////////////////////////////
// The `.bc` reader and the condition classes a Simulate_cli-style driver hands to
// LinearElasticity::Simulator::applyBoundaryConditions, with the reference's names (BoundaryConditions.hh:37-447,
// BoundaryConditions.cc:188-389, Geometry.hh:40-300, ComponentMask.hh):
//
//     conds = readBoundaryConditions<N>(path, bbox, noRigidMotion[, pps, pinTranslationComponents]);
//     sim.applyBoundaryConditions(conds);
//
// Region kinds: "box", "box%" (relative to `bbox`), "path" (polyline, 1e-5 tube), "polygon" (2D point-in-polygon);
// condition types: dirichlet[xyz] / traction / force / pressure / delta force on a region, numeric or expression valued;
// "dirichlet nodes" / "delta force nodes" lists; "traction | pressure | force elements" lists; "dirichlet elements" with
// "element vertices". "target*", "contact" and "fracture" conditions parse into their classes and are ignored / refused
// by the linear-elasticity Simulator the way the reference's applyBoundaryConditions does (:936-938, :1024).
// Vectors are std::array<Real, N>; no Eigen.
#ifndef MESHFEMHIP_BOUNDARYCONDITIONS_HH
#define MESHFEMHIP_BOUNDARYCONDITIONS_HH

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <ostream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "ExpressionVector.hh"
#include "Json.hh"

namespace MeshFEMHip {

using Real = double;
template <size_t N> using VectorND = std::array<Real, N>;
template <size_t N> using IVectorND = std::array<size_t, N>;

// which of x, y, z a Dirichlet condition constrains (ComponentMask.hh:5-60)
struct ComponentMask {
    ComponentMask(const std::string &components = "") { setComponentString(components); }
    void setComponentString(const std::string &components) {
        m_bits = 0;
        for (int c = 0; c < 3; ++c) if (components.find("xyz"[c]) != std::string::npos) m_bits |= 1 << c;
        size_t cnt = 0;
        for (int c = 0; c < 3; ++c) cnt += (m_bits >> c) & 1;
        if (cnt != components.size()) throw std::runtime_error("invalid component specifier: '" + components + "'");
    }
    bool has(size_t c) const { return (m_bits >> c) & 1; }
    bool hasX() const { return has(0); }
    bool hasY() const { return has(1); }
    bool hasZ() const { return has(2); }
    size_t count(size_t dim) const {
        if (dim != 2 && dim != 3) throw std::runtime_error("Illegal dimension");
        size_t cnt = 0;
        for (size_t c = 0; c < dim; ++c) cnt += has(c);
        return cnt;
    }
    bool hasAny(size_t dim) const { return count(dim) > 0; }
    bool hasAll(size_t dim) const { return count(dim) == dim; }
    void set() { m_bits = 7; }
    void set(size_t c) { m_bits |= 1 << c; }
    void clear() { m_bits = 0; }
    void clear(size_t c) { m_bits &= ~(1 << c); }
    std::string componentString() const {
        std::string s;
        for (int c = 0; c < 3; ++c) if (has(c)) s.push_back("xyz"[c]);
        return s;
    }
    int bits(size_t dim) const { return m_bits & ((1 << dim) - 1); }          // the C ABI's compMask
private:
    int m_bits = 0;
};

// ---------------------------------------------------------------- regions (Geometry.hh:40-300)
template <size_t N>
struct Region {
    VectorND<N> minCorner{}, maxCorner{};                      // zero for non-box regions, like the reference
    virtual ~Region() {}
    virtual bool containsPoint(const VectorND<N> &p) const = 0;
    VectorND<N> dimensions() const {
        VectorND<N> d;
        for (size_t i = 0; i < N; ++i) d[i] = maxCorner[i] - minCorner[i];
        return d;
    }
};

template <size_t N>
struct BBox : public Region<N> {
    BBox() {}
    BBox(const VectorND<N> &mn, const VectorND<N> &mx) { this->minCorner = mn; this->maxCorner = mx; }
    // smallest box around a point cloud
    template <class Points> explicit BBox(const Points &pts) {
        if (pts.empty()) return;
        this->minCorner = this->maxCorner = pts[0];
        for (const auto &p : pts)
            for (size_t i = 0; i < N; ++i) {
                this->minCorner[i] = std::min(this->minCorner[i], p[i]);
                this->maxCorner[i] = std::max(this->maxCorner[i], p[i]);
            }
    }
    bool containsPoint(const VectorND<N> &p) const override {  // inclusive (Geometry.hh:276-279)
        for (size_t i = 0; i < N; ++i) if (p[i] < this->minCorner[i] || p[i] > this->maxCorner[i]) return false;
        return true;
    }
    VectorND<N> interpolatePoint(const VectorND<N> &t) const { // relative -> absolute coordinates ("box%")
        VectorND<N> p;
        for (size_t i = 0; i < N; ++i) p[i] = this->minCorner[i] + t[i] * (this->maxCorner[i] - this->minCorner[i]);
        return p;
    }
};

// points closer than 1e-5 to a polyline (Geometry.hh:68-124)
template <size_t N>
struct PathRegion : public Region<N> {
    explicit PathRegion(const std::vector<VectorND<N>> &path) : m_path(path) {}
    bool containsPoint(const VectorND<N> &p) const override {
        for (size_t k = 0; k + 1 < m_path.size(); ++k) {
            const auto &a = m_path[k], &b = m_path[k + 1];
            Real vv = 0, pv = 0;
            for (size_t i = 0; i < N; ++i) { vv += (b[i] - a[i]) * (b[i] - a[i]); pv += (p[i] - a[i]) * (b[i] - a[i]); }
            Real t = std::min(1.0, std::max(0.0, pv / vv)), d2 = 0;
            for (size_t i = 0; i < N; ++i) { Real d = p[i] - (a[i] + t * (b[i] - a[i])); d2 += d * d; }
            if (std::sqrt(d2) < 1e-5) return true;
        }
        return false;
    }
private:
    std::vector<VectorND<N>> m_path;
};

// odd number of crossings between the polygon's edges and the segment from a fixed outside point (Geometry.hh:126-191);
// only the first two coordinates enter
template <size_t N>
struct PolygonalRegion : public Region<N> {
    explicit PolygonalRegion(const std::vector<VectorND<N>> &polygon) : m_poly(polygon) {
        Real minX = polygon.empty() ? 0.0 : polygon[0][0];
        for (const auto &p : polygon) minX = std::min(minX, p[0]);
        m_outside = {minX - 1.0, 1.90588};
    }
    bool containsPoint(const VectorND<N> &p) const override {
        auto det = [](Real ux, Real uy, Real vx, Real vy) { return ux * vy - uy * vx; };
        size_t crossings = 0;
        const Real cx = m_outside[0], cy = m_outside[1], dx = p[0], dy = p[1];
        for (size_t k = 0; k < m_poly.size(); ++k) {
            const auto &a = m_poly[k], &b = m_poly[(k + 1) % m_poly.size()];
            Real x = det(cx - a[0], cy - a[1], dx - cx, dy - cy), y = det(b[0] - a[0], b[1] - a[1], a[0] - cx, a[1] - cy),
                 z = det(b[0] - a[0], b[1] - a[1], dx - cx, dy - cy);
            bool miss = std::fabs(z) < 1e-10 || x * z < 0 || x * z > z * z || y * z < 0 || y * z > z * z;
            crossings += !miss;
        }
        return crossings % 2 == 1;
    }
private:
    std::vector<VectorND<N>> m_poly;
    std::array<Real, 2> m_outside;
};

// ---------------------------------------------------------------- conditions (BoundaryConditions.hh:37-447)
template <size_t N>
struct BoundaryCondition {
    BoundaryCondition() : region(std::make_shared<BBox<N>>()) {}
    explicit BoundaryCondition(const std::shared_ptr<Region<N>> &r) : region(r) {}
    virtual ~BoundaryCondition() {}
    bool containsPoint(const VectorND<N> &p) const { return region->containsPoint(p); }
    std::shared_ptr<Region<N>> region;
};
template <size_t N> using CondPtr = std::shared_ptr<BoundaryCondition<N>>;
template <size_t N> using ConstCondPtr = std::shared_ptr<const BoundaryCondition<N>>;

enum class NeumannType { Pressure, Traction, Force };

template <size_t N>
struct NeumannCondition : public BoundaryCondition<N> {
    NeumannCondition(const std::shared_ptr<Region<N>> &r, Real p) : BoundaryCondition<N>(r), type(NeumannType::Pressure) { m_vecValue[0] = p; }
    NeumannCondition(const std::shared_ptr<Region<N>> &r, const VectorND<N> &v, NeumannType t = NeumannType::Traction)
        : BoundaryCondition<N>(r), type(t), m_vecValue(v) {}
    NeumannCondition(const std::shared_ptr<Region<N>> &r, const ExpressionVector &ev, NeumannType t = NeumannType::Traction)
        : BoundaryCondition<N>(r), type(t), m_isExpr(true), m_exprVecValue(ev) {
        if (t != NeumannType::Traction) throw std::runtime_error("Only traction supports expression vectors");
    }
    Real pressure() const { m_noExpr(); return m_vecValue[0]; }
    VectorND<N> traction() const { m_noExpr(); return m_vecValue; }
    Real pressure(const ExpressionEnvironment &) const { return pressure(); }
    VectorND<N> traction(const ExpressionEnvironment &env) const { return m_isExpr ? m_exprVecValue.template eval<N>(env) : m_vecValue; }
    NeumannType type;
private:
    void m_noExpr() const { if (m_isExpr) throw std::runtime_error("Expression-valued condition needs an environment"); }
    VectorND<N> m_vecValue{};
    bool m_isExpr = false;
    ExpressionVector m_exprVecValue;
};

template <size_t N>
struct DirichletCondition : public BoundaryCondition<N> {
    DirichletCondition(const std::shared_ptr<Region<N>> &r, const VectorND<N> &v, const ComponentMask &mask)
        : BoundaryCondition<N>(r), componentMask(mask), m_displacement(v) {}
    DirichletCondition(const std::shared_ptr<Region<N>> &r, const ExpressionVector &ev, const ComponentMask &mask)
        : BoundaryCondition<N>(r), componentMask(mask), m_isExpr(true), m_displacementExpr(ev) {}
    VectorND<N> displacement() const {
        if (m_isExpr) throw std::runtime_error("Expression-valued condition needs an environment");
        return m_displacement;
    }
    VectorND<N> displacement(const ExpressionEnvironment &env) const { return m_isExpr ? m_displacementExpr.template eval<N>(env) : m_displacement; }
    ComponentMask componentMask;
private:
    VectorND<N> m_displacement{};
    bool m_isExpr = false;
    ExpressionVector m_displacementExpr;
};

template <size_t N>
struct TargetCondition : public DirichletCondition<N> { using DirichletCondition<N>::DirichletCondition; };

template <size_t N> struct ContactCondition : public BoundaryCondition<N> { using BoundaryCondition<N>::BoundaryCondition; };
template <size_t N> struct FractureCondition : public BoundaryCondition<N> { using BoundaryCondition<N>::BoundaryCondition; };

// corner-index set of a boundary element, order-free (UnorderedTriplet.hh); 2D edges carry a 0 in the third slot
struct UnorderedTriplet {
    UnorderedTriplet(size_t a = 0, size_t b = 0, size_t c = 0) : idx{a, b, c} { std::sort(idx.begin(), idx.end()); }
    bool operator<(const UnorderedTriplet &o) const { return idx < o.idx; }
    bool operator==(const UnorderedTriplet &o) const { return idx == o.idx; }
    std::array<size_t, 3> idx;
};

template <size_t N>
struct NeumannElementsCondition : public BoundaryCondition<N> {
    struct Value {
        Value(Real p = 0.0) : type(NeumannType::Pressure) { m_val[0] = p; }
        Value(const VectorND<N> &t, NeumannType inputType = NeumannType::Traction) : type(inputType), m_val(t) {}
        Real pressure() const { return m_val[0]; }
        VectorND<N> traction() const { return m_val; }
        VectorND<N> force() const { return m_val; }
        NeumannType type;
    private:
        VectorND<N> m_val{};
    };
    NeumannElementsCondition(NeumannType type, const std::vector<UnorderedTriplet> &corners, const std::vector<VectorND<N>> &values) {
        for (size_t i = 0; i < corners.size(); ++i)
            m_vals[corners[i]] = type == NeumannType::Pressure ? Value(values[i][0]) : Value(values[i], type);
    }
    bool hasValueForElement(const UnorderedTriplet &e) const { return m_vals.count(e) > 0; }
    const Value &getValue(const UnorderedTriplet &e) const { return m_vals.at(e); }
    size_t numElements() const { return m_vals.size(); }
private:
    std::map<UnorderedTriplet, Value> m_vals;
};

template <size_t N>
struct DirichletNodesCondition : public BoundaryCondition<N> {
    DirichletNodesCondition(const std::vector<size_t> &idx, const std::vector<VectorND<N>> &disp, const ComponentMask &mask)
        : indices(idx), displacements(disp), componentMask(mask) {}
    std::vector<size_t> indices;
    std::vector<VectorND<N>> displacements;
    ComponentMask componentMask;
};
template <size_t N>
struct TargetNodesCondition : public DirichletNodesCondition<N> { using DirichletNodesCondition<N>::DirichletNodesCondition; };

template <size_t N>
struct DirichletElementsCondition : public BoundaryCondition<N> {
    DirichletElementsCondition(const std::vector<IVectorND<N>> &corners, const VectorND<N> &v, const ComponentMask &mask)
        : componentMask(mask), m_corners(corners), m_displacement(v) { sortIndices(); }
    DirichletElementsCondition(const std::vector<IVectorND<N>> &corners, const ExpressionVector &ev, const ComponentMask &mask)
        : componentMask(mask), m_corners(corners), m_isExpr(true), m_displacementExpr(ev) { sortIndices(); }
    bool containsElement(IVectorND<N> idx) const {
        std::sort(idx.begin(), idx.end());
        return std::binary_search(m_corners.begin(), m_corners.end(), idx);
    }
    VectorND<N> displacement(const ExpressionEnvironment &env) const { return m_isExpr ? m_displacementExpr.template eval<N>(env) : m_displacement; }
    void sortIndices() {
        for (auto &idx : m_corners) std::sort(idx.begin(), idx.end());
        std::sort(m_corners.begin(), m_corners.end());
    }
    ComponentMask componentMask;
private:
    std::vector<IVectorND<N>> m_corners;
    VectorND<N> m_displacement{};
    bool m_isExpr = false;
    ExpressionVector m_displacementExpr;
};

template <size_t N>
struct DeltaForceCondition : public BoundaryCondition<N> {
    DeltaForceCondition(const std::shared_ptr<Region<N>> &r, const VectorND<N> &f) : BoundaryCondition<N>(r), m_force(f) {}
    DeltaForceCondition(const std::shared_ptr<Region<N>> &r, const ExpressionVector &ev) : BoundaryCondition<N>(r), m_isExpr(true), m_forceExpr(ev) {}
    VectorND<N> force(const ExpressionEnvironment &env) const { return m_isExpr ? m_forceExpr.template eval<N>(env) : m_force; }
private:
    VectorND<N> m_force{};
    bool m_isExpr = false;
    ExpressionVector m_forceExpr;
};

template <size_t N>
struct DeltaForceNodesCondition : public BoundaryCondition<N> {
    DeltaForceNodesCondition(const std::vector<size_t> &idx, const std::vector<VectorND<N>> &f) : indices(idx), forces(f) {}
    std::vector<size_t> indices;
    std::vector<VectorND<N>> forces;
};

// "fix_periodic_pair_<component>": "<orthogonal axis>" (BoundaryConditions.hh:54-100): component c of ONE matching pair of
// boundary nodes on the min / max faces of the axis is fixed to zero. `pair` takes the boundary nodes' positions.
template <size_t N>
class PeriodicPairDirichletCondition {
public:
    PeriodicPairDirichletCondition(size_t c, size_t f) : m_faceSpecifier(f) { m_component.set(c); }
    const ComponentMask &component() const { return m_component; }
    size_t faceSpecifier() const { return m_faceSpecifier; }
    bool hasCondition() const { return m_component.hasAny(N); }
    // indices into `bdryNodePos` (boundary-node numbering)
    std::pair<size_t, size_t> pair(const std::vector<VectorND<N>> &bdryNodePos, const BBox<N> &bbox, Real epsilon = 1e-5) {
        if (m_cached) return m_pair;
        const size_t f = m_faceSpecifier, nb = bdryNodePos.size();
        VectorND<N> pointToMatch{};
        size_t i;
        for (i = 0; i < nb; ++i)
            if (std::fabs(bdryNodePos[i][f] - bbox.minCorner[f]) <= epsilon) {
                pointToMatch = bdryNodePos[i];
                pointToMatch[f] = bbox.maxCorner[f];
                m_pair.first = i;
                break;
            }
        if (i == nb) throw std::runtime_error("No vertices on the periodic pair face.");
        for (i = 0; i < nb; ++i) {
            Real d2 = 0;
            for (size_t c = 0; c < N; ++c) d2 += (bdryNodePos[i][c] - pointToMatch[c]) * (bdryNodePos[i][c] - pointToMatch[c]);
            if (std::sqrt(d2) <= epsilon) { m_pair.second = i; break; }
        }
        if (i == nb) throw std::runtime_error("Couldn't match vertex in periodic pair Dirichlet condition");
        m_cached = true;
        return m_pair;
    }
private:
    ComponentMask m_component;
    size_t m_faceSpecifier;
    bool m_cached = false;
    std::pair<size_t, size_t> m_pair{0, 0};
};

// ---------------------------------------------------------------- the reader (BoundaryConditions.cc:25-389)
namespace detail {

// 2- or 3-vectors, padded with zeros (parseVectorLenient, :28-45)
inline std::array<Real, 3> parseVectorLenient(const Json &params) {
    std::array<Real, 3> v{0, 0, 0};
    int nRead = 0;
    if (params.is_array())
        for (const auto &val : params.items()) {
            if (!val.is_number()) { nRead = -1; break; }
            if (nRead < 3) v[(size_t)nRead] = val.number();
            ++nRead;
        }
    if (nRead != 2 && nRead != 3) throw std::runtime_error("Error parsing vector; read " + std::to_string(nRead) + " components");
    return v;
}

template <size_t N> VectorND<N> truncateFrom3D(const std::array<Real, 3> &v) {
    VectorND<N> out;
    for (size_t i = 0; i < N; ++i) out[i] = v[i];
    return out;
}

inline std::string numberString(double x) {
    char buf[64];
    snprintf(buf, sizeof(buf), "%.17g", x);
    return buf;
}

inline std::vector<std::string> parseExpressionVector(const Json &params) {     // :48-61
    std::vector<std::string> result;
    if (!params.is_array()) throw std::runtime_error("Failed to parse expression vector");
    for (const auto &val : params.items()) {
        if (val.is_string()) result.push_back(val.string());
        else if (val.is_number()) result.push_back(numberString(val.number()));
        else throw std::runtime_error("Failed to parse expression vector");
    }
    return result;
}

inline size_t parseIndex(const Json &v, const char *msg) {
    if (!v.is_number() || v.number() < 0 || v.number() != std::floor(v.number())) throw std::runtime_error(msg);
    return (size_t)v.number();
}

// [[value, [node, ...]], ...] (:63-82)
template <size_t N>
void parseNodeConditionValues(const Json &params, std::vector<size_t> &indices, std::vector<VectorND<N>> &values) {
    indices.clear(); values.clear();
    for (const auto &val : params.items()) {
        auto v = parseVectorLenient(val[0]);
        for (const auto &nd : val[1].items()) {
            indices.push_back(parseIndex(nd, "Error parsing node condition values."));
            values.push_back(truncateFrom3D<N>(v));
        }
    }
}

// [[value, [[corner, corner(, corner)], ...]], ...] (:84-112)
template <size_t N>
void parseElementConditionValues(const Json &params, std::vector<UnorderedTriplet> &corners, std::vector<VectorND<N>> &values) {
    corners.clear(); values.clear();
    const char *msg = "Error parsing element condition values.";
    for (const auto &val : params.items()) {
        auto v = parseVectorLenient(val[0]);
        for (const auto &elem : val[1].items()) {
            std::vector<size_t> idx;
            for (const auto &c : elem.items()) idx.push_back(parseIndex(c, msg));
            if (idx.size() == 2) idx.push_back(0);
            if (idx.size() != 3) throw std::runtime_error(msg);
            values.push_back(truncateFrom3D<N>(v));
            corners.emplace_back(idx[0], idx[1], idx[2]);
        }
    }
}

template <size_t N>
void parseElementVertices(const Json &params, std::vector<IVectorND<N>> &elementVertices) {   // :114-127
    elementVertices.clear();
    const char *msg = "Error parsing element vertices.";
    for (const auto &val : params.items()) {
        if (val.size() != N) throw std::runtime_error(msg);
        IVectorND<N> corners;
        for (size_t i = 0; i < N; ++i) corners[i] = parseIndex(val[i], msg);
        elementVertices.push_back(corners);
    }
}

} // namespace detail

template <size_t N>
std::vector<CondPtr<N>> readBoundaryConditions(const Json &params, const BBox<N> &bbox, bool &noRigidMotion,
                                               std::vector<PeriodicPairDirichletCondition<N>> &pps, ComponentMask &pinTranslation) {
    using namespace detail;
    std::vector<CondPtr<N>> conds;
    noRigidMotion = params.count("no_rigid_motion") ? params["no_rigid_motion"].boolean() : false;
    for (size_t c = 0; c < N; ++c) {                                 // "fix_periodic_pair_<component>": "<orthogonal axis>"
        const std::string key = std::string("fix_periodic_pair_") + "xyz"[c];
        if (!params.count(key)) continue;
        const std::string faceSpecifier = params[key].string();
        size_t face = N;
        for (size_t c2 = 0; c2 < N; ++c2)
            if (c2 != c && faceSpecifier == std::string(1, "xyz"[c2])) face = c2;
        if (face == N) throw std::runtime_error("invalid " + key);
        pps.emplace_back(c, face);
    }
    pinTranslation.setComponentString(params.count("pin_translation") ? params["pin_translation"].string() : "");

    for (const auto &tcond : params["regions"].items()) {
        std::string type = tcond["type"].string();
        std::vector<size_t> nodeIndices;
        std::vector<VectorND<N>> nodeValues;
        std::vector<IVectorND<N>> elementVertices;
        std::vector<UnorderedTriplet> elementCorners;
        std::vector<VectorND<N>> elementValues;
        std::shared_ptr<Region<N>> region = std::make_shared<BBox<N>>();
        VectorND<N> value{};
        ExpressionVector exprVec;

        // "dirichletxy", "targetz elements", ...: component letters directly after the keyword (:273-290)
        ComponentMask cmask("xyz");
        std::string prefix;
        if (type.compare(0, 9, "dirichlet") == 0) { prefix = "dirichlet"; type = type.substr(9); }
        else if (type.compare(0, 6, "target") == 0) { prefix = "target"; type = type.substr(6); }
        if (!prefix.empty()) {
            size_t len = 0;
            while (len < type.size() && type[len] >= 'x' && type[len] <= 'z') ++len;
            if (len > 3) throw std::runtime_error("invalid mask");
            if (len > 0) cmask.setComponentString(type.substr(0, len));
            type = prefix + type.substr(len);
        }

        if (type.find("nodes") != std::string::npos) parseNodeConditionValues<N>(tcond["values"], nodeIndices, nodeValues);
        else if (type == "traction elements" || type == "pressure elements" || type == "force elements")
            parseElementConditionValues<N>(tcond["values"], elementCorners, elementValues);
        else {
            auto corner = [&](const Json &box, const char *which) { return truncateFrom3D<N>(parseVectorLenient(box[which])); };
            if (tcond.count("box")) {
                region->minCorner = corner(tcond["box"], "minCorner");
                region->maxCorner = corner(tcond["box"], "maxCorner");
            } else if (tcond.count("box%")) {
                region->minCorner = bbox.interpolatePoint(corner(tcond["box%"], "minCorner"));
                region->maxCorner = bbox.interpolatePoint(corner(tcond["box%"], "maxCorner"));
            } else if (tcond.count("element vertices")) {
                parseElementVertices<N>(tcond["element vertices"], elementVertices);
            } else if (tcond.count("path")) {
                std::vector<VectorND<N>> path;
                for (const auto &pt : tcond["path"].items()) path.push_back(truncateFrom3D<N>(parseVectorLenient(pt)));
                region = std::make_shared<PathRegion<N>>(path);
            } else if (tcond.count("polygon")) {
                std::vector<VectorND<N>> polygon;
                for (const auto &pt : tcond["polygon"].items()) polygon.push_back(truncateFrom3D<N>(parseVectorLenient(pt)));
                region = std::make_shared<PolygonalRegion<N>>(polygon);
            }
            bool plain = true;                                        // plain vector first, expression vector otherwise
            try { value = truncateFrom3D<N>(parseVectorLenient(tcond["value"])); }
            catch (...) { plain = false; }
            if (!plain) {
                auto expressions = parseExpressionVector(tcond["value"]);
                if (N == 2 && expressions.size() == 3 && std::stod(expressions[2]) == 0) expressions.pop_back();
                if (expressions.size() != N) throw std::runtime_error("Incorrect expression vector size");
                for (const auto &e : expressions) exprVec.add(e);
            }
        }

        CondPtr<N> c;
        if (exprVec.size() > 0) {
            if (type == "traction") c = std::make_shared<NeumannCondition<N>>(region, exprVec, NeumannType::Traction);
            else if (type == "dirichlet") c = std::make_shared<DirichletCondition<N>>(region, exprVec, cmask);
            else if (type == "dirichlet elements") c = std::make_shared<DirichletElementsCondition<N>>(elementVertices, exprVec, cmask);
            else if (type == "target") c = std::make_shared<TargetCondition<N>>(region, exprVec, cmask);
            else if (type == "delta force") c = std::make_shared<DeltaForceCondition<N>>(region, exprVec);
            else throw std::runtime_error("Only region-based traction, dirichlet, target, and delta force support expression vectors");
        } else {
            if (type == "pressure") c = std::make_shared<NeumannCondition<N>>(region, value[0]);
            else if (type == "traction") c = std::make_shared<NeumannCondition<N>>(region, value, NeumannType::Traction);
            else if (type == "force") c = std::make_shared<NeumannCondition<N>>(region, value, NeumannType::Force);
            else if (type == "dirichlet") c = std::make_shared<DirichletCondition<N>>(region, value, cmask);
            else if (type == "dirichlet elements") c = std::make_shared<DirichletElementsCondition<N>>(elementVertices, value, cmask);
            else if (type == "target") c = std::make_shared<TargetCondition<N>>(region, value, cmask);
            else if (type == "contact") c = std::make_shared<ContactCondition<N>>(region);
            else if (type == "fracture") c = std::make_shared<FractureCondition<N>>(region);
            else if (type == "dirichlet nodes") c = std::make_shared<DirichletNodesCondition<N>>(nodeIndices, nodeValues, cmask);
            else if (type == "target nodes") c = std::make_shared<TargetNodesCondition<N>>(nodeIndices, nodeValues, cmask);
            else if (type == "traction elements") c = std::make_shared<NeumannElementsCondition<N>>(NeumannType::Traction, elementCorners, elementValues);
            else if (type == "pressure elements") c = std::make_shared<NeumannElementsCondition<N>>(NeumannType::Pressure, elementCorners, elementValues);
            else if (type == "force elements") c = std::make_shared<NeumannElementsCondition<N>>(NeumannType::Force, elementCorners, elementValues);
            else if (type == "delta force") c = std::make_shared<DeltaForceCondition<N>>(region, value);
            else if (type == "delta force nodes") c = std::make_shared<DeltaForceNodesCondition<N>>(nodeIndices, nodeValues);
            else throw std::runtime_error("Invalid type '" + type + "'");
        }
        conds.push_back(c);
    }
    return conds;
}

template <size_t N>
std::vector<CondPtr<N>> readBoundaryConditions(const std::string &cpath, const BBox<N> &bbox, bool &noRigidMotion,
                                               std::vector<PeriodicPairDirichletCondition<N>> &pps, ComponentMask &pinTranslation) {
    std::ifstream probe(cpath);
    if (!probe.is_open()) throw std::runtime_error("Couldn't open BC file:" + cpath);
    probe.close();
    return readBoundaryConditions<N>(Json::parseFile(cpath), bbox, noRigidMotion, pps, pinTranslation);
}

// three-argument form (BoundaryConditions.hh: readBoundaryConditions(path, bbox, noRigidMotion))
template <size_t N>
std::vector<CondPtr<N>> readBoundaryConditions(const std::string &cpath, const BBox<N> &bbox, bool &noRigidMotion) {
    std::vector<PeriodicPairDirichletCondition<N>> pps;
    ComponentMask pin;
    return readBoundaryConditions<N>(cpath, bbox, noRigidMotion, pps, pin);
}

// box-region conditions in the 3D-compatible format of :130-186 (other kinds: "Unsupported condition type.")
template <size_t N>
void writeBoundaryConditions(std::ostream &os, const std::vector<ConstCondPtr<N>> &conds) {
    os << "{ \"regions\": [" << std::endl;
    for (size_t i = 0; i < conds.size(); ++i) {
        const auto &c = conds[i];
        if (i > 0) os << ", ";
        os << " { \"type\": \"";
        VectorND<N> value{};
        if (auto nc = dynamic_cast<const NeumannCondition<N> *>(c.get())) {
            if (nc->type == NeumannType::Pressure) { value[0] = nc->pressure(); os << "pressure"; }
            else { value = nc->traction(); os << (nc->type == NeumannType::Traction ? "traction" : "force"); }
        } else if (auto tc = dynamic_cast<const TargetCondition<N> *>(c.get())) { os << "target"; value = tc->displacement(); }
        else if (auto dc = dynamic_cast<const DirichletCondition<N> *>(c.get())) { os << "dirichlet"; value = dc->displacement(); }
        else throw std::runtime_error("Unsupported condition type.");
        auto third = [](const VectorND<N> &v) { return N == 2 ? 0.0 : v[N - 1]; };
        os << "\", \"value\": [" << value[0] << ", " << value[1] << ", " << third(value) << "], \"box\": { \"minCorner\": ["
           << c->region->minCorner[0] << ", " << c->region->minCorner[1] << ", " << third(c->region->minCorner) << "], \"maxCorner\": ["
           << c->region->maxCorner[0] << ", " << c->region->maxCorner[1] << ", " << third(c->region->maxCorner) << "] } }";
    }
    os << "] }" << std::endl;
}

} // namespace MeshFEMHip

#endif /* end of include guard: MESHFEMHIP_BOUNDARYCONDITIONS_HH */
