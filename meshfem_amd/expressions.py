"""Expression-valued boundary-condition components (ExpressionVector.hh; `"value": [0, "sin(pi * x)", 0]` in .bc files).

The reference evaluates them with tinyexpr (third-party, codeplea/tinyexpr pinned at 4e8cc0067a1e in
cmake/MeshFEMDownloadExternal.cmake:74-80; not part of /root/reference). This is a small recursive-descent restatement of
tinyexpr's published grammar in its DEFAULT configuration (no TE_POW_FROM_RIGHT, no TE_NAT_LOG), evaluated on numpy arrays
so that one call covers every node / boundary element of a region:

    <list>   = <expr> {"," <expr>}
    <expr>   = <term> {("+" | "-") <term>}
    <term>   = <factor> {("*" | "/" | "%") <factor>}
    <factor> = <power> {"^" <power>}                 left-associative; -a^b == (-a)^b
    <power>  = {("-" | "+")} <base>
    <base>   = <constant> | <variable> | <function-0> ["(" ")"] | <function-1> <power>
             | <function-n> "(" <expr> {"," <expr>} ")" | "(" <list> ")"

Functions: abs acos asin atan atan2 ceil cos cosh e exp fac floor ln log (= log10) log10 ncr npr pi pow sin sinh sqrt tan
tanh. Variables come from the environment (ExpressionEnvironment: x, y, z, mesh_size_i, mesh_min_i, mesh_max_i,
region_size_i, region_min_i, region_max_i; LinearElasticity.hh:883-894, ExpressionVector.hh:49-54)."""
import math

import numpy as np


def _fac(a):
    a = np.asarray(a, dtype=np.float64)
    out = np.full(a.shape, np.nan)
    ok = a >= 0
    out[ok] = np.vectorize(lambda v: float(math.factorial(int(v))) if v <= 170 else math.inf)(a[ok]) if ok.any() else out[ok]
    return out


def _ncr(n, r):
    n, r = np.broadcast_arrays(np.asarray(n, dtype=np.float64), np.asarray(r, dtype=np.float64))
    out = np.full(n.shape, np.nan)
    ok = (n >= 0) & (r >= 0) & (n >= r)
    if ok.any():
        out[ok] = np.vectorize(lambda a, b: float(math.comb(int(a), int(b))))(n[ok], r[ok])
    return out


_FUNCS = {
    "abs": (1, np.abs), "acos": (1, np.arccos), "asin": (1, np.arcsin), "atan": (1, np.arctan), "atan2": (2, np.arctan2),
    "ceil": (1, np.ceil), "cos": (1, np.cos), "cosh": (1, np.cosh), "e": (0, lambda: math.e), "exp": (1, np.exp),
    "fac": (1, _fac), "floor": (1, np.floor), "ln": (1, np.log), "log": (1, np.log10), "log10": (1, np.log10),
    "ncr": (2, _ncr), "npr": (2, lambda n, r: _ncr(n, r) * _fac(r)), "pi": (0, lambda: math.pi), "pow": (2, np.power),
    "sin": (1, np.sin), "sinh": (1, np.sinh), "sqrt": (1, np.sqrt), "tan": (1, np.tan), "tanh": (1, np.tanh),
}


class ExpressionError(RuntimeError):
    pass


class Expression:
    """Compiled once per string; `eval(env)` with env: name -> scalar or array (arrays broadcast)."""

    def __init__(self, text):
        self.text = str(text)
        self._tok = self._tokenize(self.text)
        self._pos = 0
        self._fn = self._list()
        if self._peek()[0] != "end":
            raise ExpressionError("Failed to parse expression '%s'" % self.text)

    # ---- tokens
    @staticmethod
    def _tokenize(s):
        out, i, n = [], 0, len(s)
        while i < n:
            ch = s[i]
            if ch.isspace():
                i += 1
            elif ch.isdigit() or ch == ".":
                j = i
                while j < n and (s[j].isdigit() or s[j] == "."):
                    j += 1
                if j < n and s[j] in "eE":                      # strtod: optional exponent
                    k = j + 1
                    if k < n and s[k] in "+-":
                        k += 1
                    if k < n and s[k].isdigit():
                        while k < n and s[k].isdigit():
                            k += 1
                        j = k
                try:
                    out.append(("num", float(s[i:j])))
                except ValueError:
                    raise ExpressionError("Failed to parse expression '%s'" % s)
                i = j
            elif ch.isalpha() or ch == "_":
                j = i
                while j < n and (s[j].isalnum() or s[j] == "_"):
                    j += 1
                out.append(("id", s[i:j]))
                i = j
            elif ch in "+-*/^%(),":
                out.append((ch, ch))
                i += 1
            else:
                raise ExpressionError("Failed to parse expression '%s'" % s)
        out.append(("end", None))
        return out

    def _peek(self):
        return self._tok[self._pos]

    def _next(self):
        t = self._tok[self._pos]
        self._pos += 1
        return t

    def _fail(self):
        raise ExpressionError("Failed to parse expression '%s'" % self.text)

    # ---- grammar (each returns a closure env -> value)
    def _list(self):
        f = self._expr()
        while self._peek()[0] == ",":
            self._next()
            g = self._expr()
            f = (lambda a, b: lambda env: (a(env), b(env))[1])(f, g)     # comma: value of the last expression
        return f

    def _expr(self):
        f = self._term()
        while self._peek()[0] in ("+", "-"):
            op = self._next()[0]
            g = self._term()
            f = (lambda a, b, o: (lambda env: a(env) + b(env)) if o == "+" else (lambda env: a(env) - b(env)))(f, g, op)
        return f

    def _term(self):
        f = self._factor()
        while self._peek()[0] in ("*", "/", "%"):
            op = self._next()[0]
            g = self._factor()
            if op == "*":
                f = (lambda a, b: lambda env: a(env) * b(env))(f, g)
            elif op == "/":
                f = (lambda a, b: lambda env: np.divide(a(env), b(env)))(f, g)
            else:
                f = (lambda a, b: lambda env: np.fmod(a(env), b(env)))(f, g)
        return f

    def _factor(self):
        f = self._power()
        while self._peek()[0] == "^":
            self._next()
            g = self._power()
            f = (lambda a, b: lambda env: np.power(np.asarray(a(env), dtype=np.float64), b(env)))(f, g)
        return f

    def _power(self):
        sign = 1
        while self._peek()[0] in ("+", "-"):
            if self._next()[0] == "-":
                sign = -sign
        f = self._base()
        return f if sign == 1 else (lambda a: lambda env: -a(env))(f)

    def _base(self):
        kind, val = self._next()
        if kind == "num":
            return lambda env, v=val: v
        if kind == "(":
            f = self._list()
            if self._next()[0] != ")":
                self._fail()
            return f
        if kind != "id":
            self._fail()
        if val in _FUNCS:
            arity, fn = _FUNCS[val]
            if arity == 0:
                if self._peek()[0] == "(":
                    self._next()
                    if self._next()[0] != ")":
                        self._fail()
                return lambda env, fn=fn: fn()
            if arity == 1:
                g = self._power()
                return lambda env, fn=fn, g=g: fn(np.asarray(g(env), dtype=np.float64))
            if self._next()[0] != "(":
                self._fail()
            args = [self._expr()]
            while self._peek()[0] == ",":
                self._next()
                args.append(self._expr())
            if self._next()[0] != ")" or len(args) != arity:
                self._fail()
            return lambda env, fn=fn, args=args: fn(*[np.asarray(a(env), dtype=np.float64) for a in args])
        name = val

        def var(env, name=name):
            if name not in env:
                raise ExpressionError("Failed to parse expression '%s'" % self.text)   # unknown identifier (te_compile error)
            return env[name]
        return var

    def eval(self, env):
        with np.errstate(all="ignore"):
            return self._fn(env)


class ExpressionVector:
    """ExpressionVector (ExpressionVector.hh:115-140): one expression per component."""

    def __init__(self, components):
        self.exprs = [Expression(c if isinstance(c, str) else repr(float(c))) for c in components]

    def __len__(self):
        return len(self.exprs)

    def eval(self, env, npts):
        return np.column_stack([np.broadcast_to(np.asarray(e.eval(env), dtype=np.float64), (npts,)) for e in self.exprs])


def environment(dim, mesh_min, mesh_max, region_min=None, region_max=None, points=None):
    """ExpressionEnvironment as applyBoundaryConditions fills it (LinearElasticity.hh:883-894, :908/:944)."""
    env = {}

    def vec(name, v):
        for i in range(dim):
            env["%s%d" % (name, i)] = float(v[i])
    mesh_min, mesh_max = np.asarray(mesh_min, dtype=np.float64), np.asarray(mesh_max, dtype=np.float64)
    vec("mesh_size_", mesh_max - mesh_min); vec("mesh_min_", mesh_min); vec("mesh_max_", mesh_max)
    if region_min is not None:
        region_min, region_max = np.asarray(region_min, dtype=np.float64), np.asarray(region_max, dtype=np.float64)
        vec("region_size_", region_max - region_min); vec("region_min_", region_min); vec("region_max_", region_max)
    if points is not None:
        points = np.asarray(points, dtype=np.float64)
        env["x"], env["y"] = points[:, 0], points[:, 1]
        env["z"] = points[:, 2] if dim == 3 else 0.0
    return env
