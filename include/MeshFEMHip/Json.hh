////////////////////////////////////////////////////////////////////////////////
// MeshFEMHip/Json.hh
////////////////////////////////////////////////////////////////////////////////
// The JSON reader behind the `.bc` and `.material` files (the reference parses them with nlohmann::json, a third-party
// header that is not part of this image: BoundaryConditions.cc:188-226, Materials.cc:178-311). Only what those two file
// formats use: objects, arrays, strings (with the standard escapes), numbers, true / false / null. `count`, `at`,
// `operator[]`, `size`, `is_*`, `get<double>` keep nlohmann's names so that the readers above read like the reference's.
#ifndef MESHFEMHIP_JSON_HH
#define MESHFEMHIP_JSON_HH

#include <cmath>
#include <charconv>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace MeshFEMHip {

class Json {
public:
    enum class Kind { Null, Bool, Number, String, Array, Object };

    Json() = default;
    static Json parse(const std::string &text) {
        Parser p{text, 0};
        Json v = p.value();
        p.skip();
        if (p.pos != text.size()) p.fail("trailing characters");
        return v;
    }
    static Json parseFile(const std::string &path) {
        std::ifstream is(path);
        if (!is.is_open()) throw std::runtime_error("Couldn't open " + path);
        std::stringstream ss;
        ss << is.rdbuf();
        return parse(ss.str());
    }

    Kind kind() const { return m_kind; }
    bool is_null() const { return m_kind == Kind::Null; }
    bool is_boolean() const { return m_kind == Kind::Bool; }
    bool is_number() const { return m_kind == Kind::Number; }
    bool is_string() const { return m_kind == Kind::String; }
    bool is_array() const { return m_kind == Kind::Array; }
    bool is_object() const { return m_kind == Kind::Object; }

    size_t size() const { return is_array() ? m_items.size() : is_object() ? m_members.size() : (is_null() ? 0 : 1); }
    size_t count(const std::string &key) const {
        if (!is_object()) return 0;
        for (const auto &m : m_members) if (m.first == key) return 1;
        return 0;
    }
    const Json &at(const std::string &key) const {
        if (is_object())
            for (const auto &m : m_members) if (m.first == key) return m.second;
        throw std::runtime_error("key '" + key + "' not found");
    }
    const Json &operator[](const std::string &key) const { return at(key); }
    const Json &operator[](const char *key) const { return at(key); }
    const Json &at(size_t i) const {
        if (!is_array() || i >= m_items.size()) throw std::runtime_error("array index out of range");
        return m_items[i];
    }
    const Json &operator[](size_t i) const { return at(i); }
    const Json &operator[](int i) const { return at((size_t)i); }
    const std::vector<Json> &items() const { return m_items; }
    const std::vector<std::pair<std::string, Json>> &members() const { return m_members; }

    double number() const {
        if (!is_number()) throw std::runtime_error("type must be number");
        return m_num;
    }
    bool boolean() const {
        if (!is_boolean()) throw std::runtime_error("type must be boolean");
        return m_bool;
    }
    const std::string &string() const {
        if (!is_string()) throw std::runtime_error("type must be string");
        return m_str;
    }
    template <class T> T get() const;

private:
    Kind m_kind = Kind::Null;
    bool m_bool = false;
    double m_num = 0;
    std::string m_str;
    std::vector<Json> m_items;
    std::vector<std::pair<std::string, Json>> m_members;     // file order

    struct Parser {
        const std::string &s;
        size_t pos;
        [[noreturn]] void fail(const char *what) const {
            throw std::runtime_error("JSON parse error at offset " + std::to_string(pos) + ": " + what);
        }
        void skip() {
            while (pos < s.size() && (s[pos] == ' ' || s[pos] == '\t' || s[pos] == '\n' || s[pos] == '\r')) ++pos;
        }
        bool literal(const char *w) {
            size_t n = 0;
            while (w[n]) ++n;
            if (s.compare(pos, n, w) != 0) return false;
            pos += n;
            return true;
        }
        std::string str() {
            std::string out;
            ++pos;                                                       // opening quote
            while (true) {
                if (pos >= s.size()) fail("unterminated string");
                char c = s[pos++];
                if (c == '"') break;
                if (c != '\\') { out.push_back(c); continue; }
                if (pos >= s.size()) fail("unterminated escape");
                char e = s[pos++];
                switch (e) {
                    case '"': out.push_back('"'); break;
                    case '\\': out.push_back('\\'); break;
                    case '/': out.push_back('/'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'n': out.push_back('\n'); break;
                    case 'r': out.push_back('\r'); break;
                    case 't': out.push_back('\t'); break;
                    case 'u': {
                        if (pos + 4 > s.size()) fail("bad \\u escape");
                        unsigned cp = (unsigned)std::strtoul(s.substr(pos, 4).c_str(), nullptr, 16);
                        pos += 4;
                        if (cp < 0x80) out.push_back((char)cp);        // UTF-8 of the BMP code point
                        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        else { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
                        break;
                    }
                    default: fail("bad escape");
                }
            }
            return out;
        }
        Json value() {
            skip();
            if (pos >= s.size()) fail("unexpected end");
            Json v;
            char c = s[pos];
            if (c == '{') {
                v.m_kind = Kind::Object;
                ++pos; skip();
                if (pos < s.size() && s[pos] == '}') { ++pos; return v; }
                while (true) {
                    skip();
                    if (pos >= s.size() || s[pos] != '"') fail("expected a member name");
                    std::string key = str();
                    skip();
                    if (pos >= s.size() || s[pos] != ':') fail("expected ':'");
                    ++pos;
                    Json member = value();
                    bool replaced = false;
                    for (auto &m : v.m_members) if (m.first == key) { m.second = member; replaced = true; }
                    if (!replaced) v.m_members.emplace_back(std::move(key), std::move(member));
                    skip();
                    if (pos < s.size() && s[pos] == ',') { ++pos; continue; }
                    if (pos < s.size() && s[pos] == '}') { ++pos; break; }
                    fail("expected ',' or '}'");
                }
            } else if (c == '[') {
                v.m_kind = Kind::Array;
                ++pos; skip();
                if (pos < s.size() && s[pos] == ']') { ++pos; return v; }
                while (true) {
                    v.m_items.push_back(value());
                    skip();
                    if (pos < s.size() && s[pos] == ',') { ++pos; continue; }
                    if (pos < s.size() && s[pos] == ']') { ++pos; break; }
                    fail("expected ',' or ']'");
                }
            } else if (c == '"') {
                v.m_kind = Kind::String;
                v.m_str = str();
            } else if (literal("true")) { v.m_kind = Kind::Bool; v.m_bool = true; }
            else if (literal("false")) { v.m_kind = Kind::Bool; v.m_bool = false; }
            else if (literal("null")) { v.m_kind = Kind::Null; }
            else {
                // the JSON number grammar, -?(0|[1-9][0-9]*)(\.[0-9]+)?([eE][+-]?[0-9]+)?, checked BEFORE the conversion: a bare strtod also takes
                // nan / inf / hex floats / a leading '+' or '.', which nlohmann::json (the reference's reader) rejects, and reads the decimal
                // point of the process's LC_NUMERIC; std::from_chars is locale-independent
                size_t q = pos;
                auto digit = [&](size_t i) { return i < s.size() && s[i] >= '0' && s[i] <= '9'; };
                if (q < s.size() && s[q] == '-') ++q;
                if (!digit(q)) fail("unexpected character");
                if (s[q] == '0') ++q; else while (digit(q)) ++q;
                if (q < s.size() && s[q] == '.') { ++q; if (!digit(q)) fail("invalid number: digits expected after the decimal point"); while (digit(q)) ++q; }
                if (q < s.size() && (s[q] == 'e' || s[q] == 'E')) {
                    ++q;
                    if (q < s.size() && (s[q] == '+' || s[q] == '-')) ++q;
                    if (!digit(q)) fail("invalid number: digits expected in the exponent");
                    while (digit(q)) ++q;
                }
                double x = 0;
                const auto res = std::from_chars(s.data() + pos, s.data() + q, x);
                if (res.ec == std::errc::invalid_argument || res.ptr != s.data() + q) fail("invalid number");
                // (out of range: from_chars leaves x unmodified; like nlohmann, overflow to infinity is an error, underflow to zero is not)
                if (res.ec == std::errc::result_out_of_range) { x = std::strtod(std::string(s, pos, q - pos).c_str(), nullptr); if (!(x == x) || x > 1.7e308 || x < -1.7e308) fail("number out of range"); }
                v.m_kind = Kind::Number;
                v.m_num = x;
                pos = q;
            }
            return v;
        }
    };
};

template <> inline double Json::get<double>() const { return number(); }
template <> inline bool Json::get<bool>() const { return boolean(); }
template <> inline std::string Json::get<std::string>() const { return string(); }
template <> inline size_t Json::get<size_t>() const { return (size_t)number(); }
template <> inline int Json::get<int>() const { return (int)number(); }

} // namespace MeshFEMHip

#endif /* end of include guard: MESHFEMHIP_JSON_HH */
