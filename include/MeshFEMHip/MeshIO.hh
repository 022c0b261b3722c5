////////////////////////////////////////////////////////////////////////////////
// MeshFEMHip/MeshIO.hh
////////////////////////////////////////////////////////////////////////////////
// Gmsh MSH 2.2 input / output for the C++ facade, the format Simulate_cli reads and writes: MeshIO::load (the MSH branch
// of MeshIO.cc:525-760 -- ASCII and binary, consecutively numbered 1-indexed nodes, ONE element type per file: tri = 2,
// tet = 4, tri6 = 9, tet10 = 11) and MSHFieldWriter (MSHFieldWriter.hh:40-310: per-node and per-element scalar / vector /
// symmetric-matrix fields; 2-vectors are padded to 3, symmetric matrices go out as padded 3x3 scanlines). The same
// format meshfem_amd/mesh_io.py implements for the Python driver; files written by either side load in the other.
#ifndef MESHFEMHIP_MESHIO_HH
#define MESHFEMHIP_MESHIO_HH

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace MeshFEMHip {
namespace MeshIO {

struct IOVertex { std::array<double, 3> point{0, 0, 0}; double operator[](size_t i) const { return point[i]; } double &operator[](size_t i) { return point[i]; } };
using IOElement = std::vector<size_t>;                          // vertex (node) indices, 0-based
enum class MeshType { TRI, TET, INVALID };

struct FieldData { std::string domain; size_t dim = 0; std::vector<double> values; };   // domain: "node" | "element"

namespace detail {
inline size_t nodesForGmshType(int t) {                        // MeshIO.cc:527-531
    switch (t) { case 1: return 2; case 2: return 3; case 3: return 4; case 4: return 4; case 5: return 8; case 8: return 3; case 9: return 6; case 11: return 10; }
    throw std::runtime_error("Unsupported MSH element type " + std::to_string(t));
}
inline int gmshTypeForNodes(size_t n) {
    switch (n) { case 3: return 2; case 4: return 4; case 6: return 9; case 10: return 11; }
    throw std::runtime_error("Unsupported element size " + std::to_string(n));
}
inline std::string nextLine(std::istream &is) {               // next non-blank line, trimmed
    std::string line;
    while (std::getline(is, line)) {
        size_t b = line.find_first_not_of(" \t\r\n"), e = line.find_last_not_of(" \t\r\n");
        if (b != std::string::npos) return line.substr(b, e - b + 1);
    }
    return "";
}
template <class T> T readRaw(std::istream &is) {
    T v;
    is.read(reinterpret_cast<char *>(&v), sizeof(T));
    if (!is) throw std::runtime_error("Bad MSH file format");
    return v;
}
} // namespace detail

// Reads nodes, elements and (optionally) the $NodeData / $ElementData fields. Returns the mesh type.
inline MeshType load_msh(const std::string &path, std::vector<IOVertex> &nodes, std::vector<IOElement> &elements,
                         std::map<std::string, FieldData> *fields = nullptr) {
    using namespace detail;
    std::ifstream is(path, std::ios::binary);
    if (!is.is_open()) throw std::runtime_error("Couldn't open input file " + path);
    const std::runtime_error bad("Bad MSH file format"), unsupported("Unsupported MSH file format");
    if (nextLine(is) != "$MeshFormat") throw bad;
    std::istringstream fmt(nextLine(is));
    std::string version;
    int fileType = -1, dataSize = 0;
    fmt >> version >> fileType >> dataSize;
    if (fileType < 0 || fileType > 1 || dataSize != 8) throw unsupported;
    const bool binary = fileType == 1;
    if (binary && readRaw<int32_t>(is) != 1) throw unsupported;             // endianness marker
    if (nextLine(is) != "$EndMeshFormat" || nextLine(is) != "$Nodes") throw bad;
    size_t nn = std::stoul(nextLine(is));
    nodes.assign(nn, IOVertex());
    for (size_t i = 0; i < nn; ++i) {
        long idx;
        if (binary) { idx = readRaw<int32_t>(is); for (int c = 0; c < 3; ++c) nodes[i][c] = readRaw<double>(is); }
        else { std::istringstream ls(nextLine(is)); ls >> idx >> nodes[i][0] >> nodes[i][1] >> nodes[i][2]; if (!ls) throw bad; }
        if (idx != (long)i + 1) throw unsupported;                           // consecutive, 1-indexed
    }
    if (nextLine(is) != "$EndNodes" || nextLine(is) != "$Elements") throw bad;
    size_t ne = std::stoul(nextLine(is));
    elements.clear();
    elements.reserve(ne);
    int etype = -1;
    if (binary) {
        size_t read = 0;
        while (read < ne) {
            int32_t t = readRaw<int32_t>(is), cnt = readRaw<int32_t>(is), ntags = readRaw<int32_t>(is);
            if (etype < 0) etype = t;
            if (t != etype || cnt <= 0) throw bad;
            size_t k = nodesForGmshType(t);
            for (int32_t e = 0; e < cnt; ++e) {
                readRaw<int32_t>(is);
                for (int32_t g = 0; g < ntags; ++g) readRaw<int32_t>(is);
                IOElement el(k);
                for (size_t c = 0; c < k; ++c) el[c] = (size_t)(readRaw<int32_t>(is) - 1);
                elements.push_back(std::move(el));
            }
            read += (size_t)cnt;
        }
    } else {
        for (size_t e = 0; e < ne; ++e) {
            std::istringstream ls(nextLine(is));
            long idx, t, ntags, tag;
            ls >> idx >> t >> ntags;
            if (!ls) throw bad;
            if (etype < 0) etype = (int)t;
            if (t != etype) throw bad;
            for (long g = 0; g < ntags; ++g) ls >> tag;
            size_t k = nodesForGmshType((int)t);
            IOElement el(k);
            for (size_t c = 0; c < k; ++c) { long v; ls >> v; el[c] = (size_t)(v - 1); }
            if (!ls) throw bad;
            elements.push_back(std::move(el));
        }
    }
    if (nextLine(is) != "$EndElements") throw bad;
    if (fields) {
        fields->clear();
        while (true) {
            std::string hdr = nextLine(is);
            if (hdr.empty()) break;
            if (hdr != "$NodeData" && hdr != "$ElementData") {
                if (hdr == "$ElementNodeData") {                          // skipped: not needed by the drivers
                    std::string end;
                    while (!(end = nextLine(is)).empty() && end != "$EndElementNodeData") {}
                }
                continue;
            }
            size_t nstr = std::stoul(nextLine(is));
            std::string name;
            for (size_t s = 0; s < nstr; ++s) { std::string t = nextLine(is); if (s == 0) name = t.substr(1, t.size() - 2); }
            size_t nreal = std::stoul(nextLine(is));
            for (size_t s = 0; s < nreal; ++s) nextLine(is);
            size_t nint = std::stoul(nextLine(is));
            std::vector<long> itags(nint);
            for (size_t s = 0; s < nint; ++s) itags[s] = std::stol(nextLine(is));
            if (nint < 3) throw bad;
            FieldData fd;
            fd.domain = hdr == "$NodeData" ? "node" : "element";
            fd.dim = (size_t)itags[1];
            size_t cnt = (size_t)itags[2];
            fd.values.resize(cnt * fd.dim);
            for (size_t i = 0; i < cnt; ++i) {
                if (binary) { readRaw<int32_t>(is); for (size_t c = 0; c < fd.dim; ++c) fd.values[i * fd.dim + c] = readRaw<double>(is); }
                else { std::istringstream ls(nextLine(is)); long idx; ls >> idx; for (size_t c = 0; c < fd.dim; ++c) ls >> fd.values[i * fd.dim + c]; if (!ls) throw bad; }
            }
            nextLine(is);                                                 // $End...
            (*fields)[name] = std::move(fd);
        }
    }
    if (elements.empty()) return MeshType::INVALID;
    size_t k = elements[0].size();
    return (k == 3 || k == 6) ? MeshType::TRI : (k == 4 || k == 10) ? MeshType::TET : MeshType::INVALID;
}

// MeshIO::load(path, nodes, elements): the format is picked from the extension (only .msh in the C++ facade; the Python
// driver also reads .off / .obj / .mesh)
inline MeshType load(const std::string &path, std::vector<IOVertex> &nodes, std::vector<IOElement> &elements) {
    size_t dot = path.rfind('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    if (ext != ".msh") throw std::runtime_error("Unsupported mesh format '" + ext + "' (the C++ facade reads .msh)");
    return load_msh(path, nodes, elements);
}

} // namespace MeshIO

// MSHFieldWriter (MSHFieldWriter.hh): mesh in the constructor, then addField calls; binary by default like the reference
class MSHFieldWriter {
public:
    enum class Domain { PER_NODE, PER_ELEMENT };
    template <class Nodes, class Elements>
    MSHFieldWriter(const std::string &path, const Nodes &nodes, const Elements &elements, bool binary = true)
        : m_os(path, std::ios::binary), m_binary(binary), m_numNodes(nodes.size()), m_numElements(elements.size()) {
        if (!m_os.is_open()) throw std::runtime_error("Couldn't open output file " + path);
        m_os << "$MeshFormat\n2.2 " << (binary ? 1 : 0) << " 8\n";
        if (binary) { m_raw<int32_t>(1); m_os << "\n"; }
        m_os << "$EndMeshFormat\n$Nodes\n" << nodes.size() << "\n";
        for (size_t i = 0; i < nodes.size(); ++i) {
            double p[3] = {0, 0, 0};
            for (size_t c = 0; c < nodes[i].size() && c < 3; ++c) p[c] = nodes[i][c];
            if (binary) { m_raw<int32_t>((int32_t)(i + 1)); for (double x : p) m_raw<double>(x); }
            else m_os << (i + 1) << " " << m_fmt(p[0]) << " " << m_fmt(p[1]) << " " << m_fmt(p[2]) << "\n";
        }
        if (binary) m_os << "\n";
        m_os << "$EndNodes\n$Elements\n" << elements.size() << "\n";
        if (elements.size()) {
            int etype = MeshIO::detail::gmshTypeForNodes(elements[0].size());
            if (binary) { m_raw<int32_t>(etype); m_raw<int32_t>((int32_t)elements.size()); m_raw<int32_t>(0); }
            for (size_t i = 0; i < elements.size(); ++i) {
                if (binary) { m_raw<int32_t>((int32_t)(i + 1)); for (size_t c = 0; c < elements[i].size(); ++c) m_raw<int32_t>((int32_t)(elements[i][c] + 1)); }
                else {
                    m_os << (i + 1) << " " << etype << " 0";
                    for (size_t c = 0; c < elements[i].size(); ++c) m_os << " " << (elements[i][c] + 1);
                    m_os << "\n";
                }
            }
        }
        if (binary) m_os << "\n";
        m_os << "$EndElements\n";
    }
    size_t numNodes() const { return m_numNodes; }
    size_t numElements() const { return m_numElements; }

    // scalar field: one value per node / element
    void addField(const std::string &name, const std::vector<double> &values, Domain domain) { m_write(name, values.data(), values.size(), 1, 1, domain); }
    // vector field: std::array<double, 2 | 3> per entry (2-vectors padded with z = 0)
    template <size_t N> void addField(const std::string &name, const std::vector<std::array<double, N>> &values, Domain domain) {
        static_assert(N == 2 || N == 3, "vector fields are 2- or 3-dimensional");
        m_write(name, &values[0][0], values.size(), N, 3, domain);
    }
    // symmetric-matrix field in the flattened order xx,yy,(zz,yz,xz,)xy -> padded 3x3 scanline (MSHFieldWriter.hh:160-170)
    template <size_t FL> void addSymmetricMatrixField(const std::string &name, const std::vector<std::array<double, FL>> &values, Domain domain) {
        static_assert(FL == 3 || FL == 6, "flattened 2x2 or 3x3 symmetric matrices");
        std::vector<double> full(values.size() * 9, 0.0);
        static const int idx2[3][2] = {{0, 0}, {1, 1}, {0, 1}}, idx3[6][2] = {{0, 0}, {1, 1}, {2, 2}, {1, 2}, {0, 2}, {0, 1}};
        for (size_t i = 0; i < values.size(); ++i)
            for (size_t q = 0; q < FL; ++q) {
                int a = FL == 3 ? idx2[q][0] : idx3[q][0], b = FL == 3 ? idx2[q][1] : idx3[q][1];
                full[i * 9 + (size_t)(a * 3 + b)] = full[i * 9 + (size_t)(b * 3 + a)] = values[i][q];
            }
        m_write(name, full.data(), values.size(), 9, 9, domain);
    }
    void close() { m_os.close(); }

private:
    template <class T> void m_raw(T v) { m_os.write(reinterpret_cast<const char *>(&v), sizeof(T)); }
    static std::string m_fmt(double x) { char b[40]; snprintf(b, sizeof(b), "%.17g", x); return b; }
    void m_write(const std::string &name, const double *v, size_t n, size_t dimIn, size_t dimOut, Domain domain) {
        if (n != (domain == Domain::PER_NODE ? m_numNodes : m_numElements)) throw std::runtime_error("Invalid field domain size.");
        const char *sec = domain == Domain::PER_NODE ? "NodeData" : "ElementData";
        m_os << "$" << sec << "\n1\n\"" << name << "\"\n0\n3\n0\n" << dimOut << "\n" << n << "\n";
        for (size_t i = 0; i < n; ++i) {
            if (m_binary) { m_raw<int32_t>((int32_t)(i + 1)); for (size_t c = 0; c < dimOut; ++c) m_raw<double>(c < dimIn ? v[i * dimIn + c] : 0.0); }
            else {
                m_os << (i + 1);
                for (size_t c = 0; c < dimOut; ++c) m_os << " " << m_fmt(c < dimIn ? v[i * dimIn + c] : 0.0);
                m_os << "\n";
            }
        }
        if (m_binary) m_os << "\n";
        m_os << "$End" << sec << "\n";
    }
    std::ofstream m_os;
    bool m_binary;
    size_t m_numNodes, m_numElements;
};

} // namespace MeshFEMHip

#endif /* end of include guard: MESHFEMHIP_MESHIO_HH */
