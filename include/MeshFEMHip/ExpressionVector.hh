////////////////////////////////////////////////////////////////////////////////
// MeshFEMHip/ExpressionVector.hh
////////////////////////////////////////////////////////////////////////////////
// Expression-valued boundary-condition components (`"value": [0, "sin(pi * x)", 0]`): ExpressionEnvironment and
// ExpressionVector with the reference's interface (ExpressionVector.hh:30-140). The reference hands the strings to tinyexpr
// (third-party, codeplea/tinyexpr pinned at 4e8cc0067a1e in cmake/MeshFEMDownloadExternal.cmake:74-80, not part of
// /root/reference); this is a recursive-descent restatement of tinyexpr's published grammar in its default configuration
// (left-associative '^', -a^b == (-a)^b, log == log10), the same grammar meshfem_amd/expressions.py implements on arrays:
//
//     <list>   = <expr> {"," <expr>}
//     <expr>   = <term> {("+" | "-") <term>}
//     <term>   = <factor> {("*" | "/" | "%") <factor>}
//     <factor> = <power> {"^" <power>}
//     <power>  = {("-" | "+")} <base>
//     <base>   = <constant> | <variable> | <function-0> ["(" ")"] | <function-1> <power>
//              | <function-n> "(" <expr> {"," <expr>} ")" | "(" <list> ")"
#ifndef MESHFEMHIP_EXPRESSIONVECTOR_HH
#define MESHFEMHIP_EXPRESSIONVECTOR_HH

#include <array>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace MeshFEMHip {

// name -> value bindings (ExpressionVector.hh:30-62)
struct ExpressionEnvironment {
    void setValue(const std::string &name, double val) { m_vars[name] = val; }
    template <class Vec> void setVectorValue(const std::string &name, const Vec &v) {        // name0, name1, ...
        for (size_t i = 0; i < v.size(); ++i) setValue(name + std::to_string(i), v[i]);
    }
    template <class Vec> void setXYZ(const Vec &p) {                                           // x, y, z (z = 0 in 2D)
        static const char *names[3] = {"x", "y", "z"};
        for (size_t i = 0; i < 3; ++i) setValue(names[i], i < p.size() ? p[i] : 0.0);
    }
    const std::map<std::string, double> &variables() const { return m_vars; }
private:
    std::map<std::string, double> m_vars;
};

class Expression {
public:
    explicit Expression(const std::string &text) : m_text(text) {
        m_tokenize();
        m_pos = 0;
        m_root = m_list();
        if (m_peek().kind != Tok::End) m_fail();
    }
    const std::string &text() const { return m_text; }
    double eval(const ExpressionEnvironment &env) const { return m_eval(*m_root, env); }

private:
    struct Tok { enum Kind { Num, Id, Op, End } kind; double num; std::string id; char op; };
    struct Node {
        enum Kind { Const, Var, Neg, Add, Sub, Mul, Div, Mod, Pow, Comma, Call } kind;
        double value = 0;
        std::string name;
        int fn = -1;
        std::vector<std::unique_ptr<Node>> kids;
    };
    using NodePtr = std::unique_ptr<Node>;

    // tinyexpr's builtin table (alphabetical there too)
    struct Fn { const char *name; int arity; };
    static const Fn *m_functions(int &count) {
        static const Fn table[] = {{"abs", 1}, {"acos", 1}, {"asin", 1}, {"atan", 1}, {"atan2", 2}, {"ceil", 1}, {"cos", 1}, {"cosh", 1},
                                   {"e", 0}, {"exp", 1}, {"fac", 1}, {"floor", 1}, {"ln", 1}, {"log", 1}, {"log10", 1}, {"ncr", 2},
                                   {"npr", 2}, {"pi", 0}, {"pow", 2}, {"sin", 1}, {"sinh", 1}, {"sqrt", 1}, {"tan", 1}, {"tanh", 1}};
        count = (int)(sizeof(table) / sizeof(table[0]));
        return table;
    }
    static int m_findFunction(const std::string &name) {
        int n;
        const Fn *t = m_functions(n);
        for (int i = 0; i < n; ++i) if (name == t[i].name) return i;
        return -1;
    }
    static double m_fac(double a) {
        if (a < 0) return std::numeric_limits<double>::quiet_NaN();
        if (a > 170) return std::numeric_limits<double>::infinity();
        double r = 1;
        for (unsigned long i = 2, n = (unsigned long)a; i <= n; ++i) r *= (double)i;
        return r;
    }
    static double m_ncr(double n, double r) {
        if (n < 0 || r < 0 || n < r) return std::numeric_limits<double>::quiet_NaN();
        unsigned long un = (unsigned long)n, ur = (unsigned long)r;
        if (ur > un / 2) ur = un - ur;
        double result = 1;
        for (unsigned long i = 1; i <= ur; ++i) { result *= (double)(un - ur + i); result /= (double)i; }
        return result;
    }
    static double m_call(int fn, const double *a) {
        int n;
        const std::string name = m_functions(n)[fn].name;
        if (name == "abs") return std::fabs(a[0]);
        if (name == "acos") return std::acos(a[0]);
        if (name == "asin") return std::asin(a[0]);
        if (name == "atan") return std::atan(a[0]);
        if (name == "atan2") return std::atan2(a[0], a[1]);
        if (name == "ceil") return std::ceil(a[0]);
        if (name == "cos") return std::cos(a[0]);
        if (name == "cosh") return std::cosh(a[0]);
        if (name == "e") return 2.71828182845904523536;
        if (name == "exp") return std::exp(a[0]);
        if (name == "fac") return m_fac(a[0]);
        if (name == "floor") return std::floor(a[0]);
        if (name == "ln") return std::log(a[0]);
        if (name == "log" || name == "log10") return std::log10(a[0]);
        if (name == "ncr") return m_ncr(a[0], a[1]);
        if (name == "npr") return m_ncr(a[0], a[1]) * m_fac(a[1]);
        if (name == "pi") return 3.14159265358979323846;
        if (name == "pow") return std::pow(a[0], a[1]);
        if (name == "sin") return std::sin(a[0]);
        if (name == "sinh") return std::sinh(a[0]);
        if (name == "sqrt") return std::sqrt(a[0]);
        if (name == "tan") return std::tan(a[0]);
        return std::tanh(a[0]);
    }

    [[noreturn]] void m_fail() const { throw std::runtime_error("Failed to parse expression '" + m_text + "'"); }

    void m_tokenize() {
        const std::string &s = m_text;
        size_t i = 0, n = s.size();
        while (i < n) {
            unsigned char ch = (unsigned char)s[i];
            if (std::isspace(ch)) { ++i; continue; }
            if (std::isdigit(ch) || ch == '.') {
                const char *b = s.c_str() + i;
                char *e = nullptr;
                double v = std::strtod(b, &e);
                if (e == b) m_fail();
                m_tok.push_back({Tok::Num, v, "", 0});
                i += (size_t)(e - b);
            } else if (std::isalpha(ch) || ch == '_') {
                size_t j = i;
                while (j < n && (std::isalnum((unsigned char)s[j]) || s[j] == '_')) ++j;
                m_tok.push_back({Tok::Id, 0, s.substr(i, j - i), 0});
                i = j;
            } else if (std::string("+-*/^%(),").find((char)ch) != std::string::npos) {
                m_tok.push_back({Tok::Op, 0, "", (char)ch});
                ++i;
            } else m_fail();
        }
        m_tok.push_back({Tok::End, 0, "", 0});
    }
    const Tok &m_peek() const { return m_tok[m_pos]; }
    const Tok &m_next() { return m_tok[m_pos++]; }
    bool m_isOp(char c) const { return m_peek().kind == Tok::Op && m_peek().op == c; }

    static NodePtr m_make(typename Node::Kind k, NodePtr a = nullptr, NodePtr b = nullptr) {
        NodePtr n(new Node);
        n->kind = k;
        if (a) n->kids.push_back(std::move(a));
        if (b) n->kids.push_back(std::move(b));
        return n;
    }
    NodePtr m_list() {
        NodePtr f = m_expr();
        while (m_isOp(',')) { m_next(); f = m_make(Node::Comma, std::move(f), m_expr()); }
        return f;
    }
    NodePtr m_expr() {
        NodePtr f = m_term();
        while (m_isOp('+') || m_isOp('-')) {
            char op = m_next().op;
            f = m_make(op == '+' ? Node::Add : Node::Sub, std::move(f), m_term());
        }
        return f;
    }
    NodePtr m_term() {
        NodePtr f = m_factor();
        while (m_isOp('*') || m_isOp('/') || m_isOp('%')) {
            char op = m_next().op;
            f = m_make(op == '*' ? Node::Mul : op == '/' ? Node::Div : Node::Mod, std::move(f), m_factor());
        }
        return f;
    }
    NodePtr m_factor() {
        NodePtr f = m_power();
        while (m_isOp('^')) { m_next(); f = m_make(Node::Pow, std::move(f), m_power()); }
        return f;
    }
    NodePtr m_power() {
        int sign = 1;
        while (m_isOp('+') || m_isOp('-')) if (m_next().op == '-') sign = -sign;
        NodePtr f = m_base();
        return sign == 1 ? std::move(f) : m_make(Node::Neg, std::move(f));
    }
    NodePtr m_base() {
        Tok t = m_next();
        if (t.kind == Tok::Num) { NodePtr n = m_make(Node::Const); n->value = t.num; return n; }
        if (t.kind == Tok::Op && t.op == '(') {
            NodePtr f = m_list();
            if (!m_isOp(')')) m_fail();
            m_next();
            return f;
        }
        if (t.kind != Tok::Id) m_fail();
        int fn = m_findFunction(t.id);
        if (fn < 0) { NodePtr n = m_make(Node::Var); n->name = t.id; return n; }
        int cnt;
        int arity = m_functions(cnt)[fn].arity;
        NodePtr n = m_make(Node::Call);
        n->fn = fn;
        if (arity == 0) {
            if (m_isOp('(')) { m_next(); if (!m_isOp(')')) m_fail(); m_next(); }
        } else if (arity == 1) {
            n->kids.push_back(m_power());
        } else {
            if (!m_isOp('(')) m_fail();
            m_next();
            n->kids.push_back(m_expr());
            while (m_isOp(',')) { m_next(); n->kids.push_back(m_expr()); }
            if (!m_isOp(')') || (int)n->kids.size() != arity) m_fail();
            m_next();
        }
        return n;
    }

    double m_eval(const Node &n, const ExpressionEnvironment &env) const {
        switch (n.kind) {
            case Node::Const: return n.value;
            case Node::Var: {
                auto it = env.variables().find(n.name);
                if (it == env.variables().end()) m_fail();           // unknown identifier: te_compile fails in the reference
                return it->second;
            }
            case Node::Neg: return -m_eval(*n.kids[0], env);
            case Node::Add: return m_eval(*n.kids[0], env) + m_eval(*n.kids[1], env);
            case Node::Sub: return m_eval(*n.kids[0], env) - m_eval(*n.kids[1], env);
            case Node::Mul: return m_eval(*n.kids[0], env) * m_eval(*n.kids[1], env);
            case Node::Div: return m_eval(*n.kids[0], env) / m_eval(*n.kids[1], env);
            case Node::Mod: return std::fmod(m_eval(*n.kids[0], env), m_eval(*n.kids[1], env));
            case Node::Pow: return std::pow(m_eval(*n.kids[0], env), m_eval(*n.kids[1], env));
            case Node::Comma: m_eval(*n.kids[0], env); return m_eval(*n.kids[1], env);
            case Node::Call: {
                double a[2] = {0, 0};
                for (size_t i = 0; i < n.kids.size() && i < 2; ++i) a[i] = m_eval(*n.kids[i], env);
                return m_call(n.fn, a);
            }
        }
        return 0;
    }

    std::string m_text;
    std::vector<Tok> m_tok;
    size_t m_pos = 0;
    NodePtr m_root;
};

// one expression per component (ExpressionVector.hh:115-140)
class ExpressionVector {
public:
    ExpressionVector() = default;
    explicit ExpressionVector(const std::vector<std::string> &components) { for (const auto &c : components) add(c); }
    void add(const std::string &expr) { m_exprs.push_back(std::make_shared<Expression>(expr)); }
    size_t size() const { return m_exprs.size(); }
    template <size_t N> std::array<double, N> eval(const ExpressionEnvironment &env) const {
        if (m_exprs.size() != N) throw std::runtime_error("Invalid evaluation size.");
        std::array<double, N> out;
        for (size_t i = 0; i < N; ++i) out[i] = m_exprs[i]->eval(env);
        return out;
    }
private:
    std::vector<std::shared_ptr<Expression>> m_exprs;
};

} // namespace MeshFEMHip

#endif /* end of include guard: MESHFEMHIP_EXPRESSIONVECTOR_HH */
