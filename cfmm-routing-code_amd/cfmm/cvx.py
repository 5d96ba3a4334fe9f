"""cfmm.cvx -- the slice of cvxpy's modelling surface the reference scripts use, mapped onto cfmm.Problem.

    import cfmm.cvx as cp          # instead of:  import cvxpy as cp

With that one line changed, /root/reference/arbitrage.py, liquidation.py and two-asset.py run as written (SURVEY 8(f)
rank 4): `cp.Variable(n, nonneg=True)`, numpy-on-the-left affine arithmetic (`A_i @ (L - D)`, `R + gamma * D - L`,
`market_value @ psi`, `psi[4]`, `psi + current_assets`), `cp.sum`, `cp.geo_mean(x, p=...)`, `>=` / `==`,
`cp.Maximize`, `cp.Problem(obj, cons).solve()`, and `.value` on variables, expressions, the objective and the problem
(arbitrage.py:51-84, liquidation.py:51-87, two-asset.py:60-100).

This is NOT a general convex modelling layer.  `Problem.solve()` pattern-matches the constraint list into the routing
problem's vocabulary and refuses anything else with `NotImplementedError`:

  * `geo_mean(R + gamma*D - L, p=w) >= geo_mean(R[, p=w])`      -> a (weighted) geometric-mean pool   arbitrage.py:65,68-70
  * `sum(R + gamma*D - L) >= sum(R)` with `R + gamma*D - L >= 0` -> a constant-sum pool               arbitrage.py:73-74
  * `sum(x) - alpha*inv_prod(x) >= sum(R) - alpha*inv_prod(R)`,  x = R + gamma*D - L (2..8 assets)  -> a stableswap pool
  * `sum(power(x, q)) >= sum(power(R, q))`, 0 < q < 1, x as above (two assets)                        -> a power-sum pool
    (neither is in the reference's scripts: they are how its pattern -- "a pool is whatever constraint line is written",
     arbitrage.py:63-74 -- extends to the library's other trading functions, in DCP-valid cvxpy)
  * rows of  psi + h (>= | ==) 0  with  psi = sum_i A_i (L_i - D_i)  and a linear objective in psi     -> the utility
    (arbitrage.py:57,77; liquidation.py:57,77-80; two-asset.py:66,86)

  * and, beyond the reference's linear objectives (include/cfmm.h: the utility table), separable concave terms on entries of psi:
    `cp.sum(cp.multiply(a, cp.log(psi + h)))` (or `a @ cp.log(psi + h)`)                              -> CFMM_ULOG entries
    `c @ psi - cp.sum(cp.multiply(k, cp.square(psi)))` (or `cp.sum_squares`)                          -> CFMM_UQUAD, depth 1 / (2 k)

and hands the result to cfmm.Problem, i.e. to libcfmm_hip.so on the MI355X -- there is no CPU path here either.
"""
import builtins
import hashlib

import numpy as np

from .problem import Problem as _RoutingProblem, Utility as _Utility
from ._lib import GE as _GE, EQ as _EQ, FREE as _FREE, ULOG as _ULOG, UQUAD as _UQUAD

# tests only: a callable n_tokens -> device context standing in for cfmm._lib.Context (the product leaves it None)
CONTEXT_FACTORY = None

OPTIMAL, INACCURATE, INFEASIBLE = "optimal", "optimal_inaccurate", "infeasible"

# Pools that stay resident.  two-asset.py:40-100 re-states the SAME five pools for each of its 50 amounts -- new variables, new
# constraints, a new cp.Problem per point -- and a conic solver starts from scratch every time.  Here the pools of the last few
# models stay uploaded (keyed by what defines them: token lists, reserves, fees, functions, weights): a model over the same pools
# only sends its utility and starts from the previous prices, as `cfmm.Problem.solve(warm_start=True)` does for the native interface.
# RESIDENT_MAX = 0 switches it off.
RESIDENT_MAX = 2
_resident = {}


def _pool_signature(n, local, pools):
    h = hashlib.blake2b(digest_size=16)
    h.update(np.int64(n).tobytes())
    for l, pl in zip(local, pools):
        h.update(np.asarray(l, dtype=np.int64).tobytes()); h.update(b"|")
        h.update(np.asarray(pl["R"], dtype=np.float64).tobytes()); h.update(np.float64(pl["fee"]).tobytes())
        h.update(str(pl["kind"]).encode())
        h.update(b"-" if pl["w"] is None else np.asarray(pl["w"], dtype=np.float64).tobytes())
        h.update(b"-" if pl.get("param") is None else np.float64(pl["param"]).tobytes())
        h.update(b";")
    return h.hexdigest(), id(CONTEXT_FACTORY)


def _const(x):
    a = np.asarray(x, dtype=np.float64)
    return a.reshape(1) if a.ndim == 0 else a


class Expression:
    """affine vector expression  const + sum_v coef[v] @ v   (1-D; a scalar is length 1 with .scalar set)"""
    __array_ufunc__ = None          # numpy on the left defers to our reflected operators
    __array_priority__ = 1000

    def __init__(self, coefs, const, scalar=False):
        self.coefs = coefs           # {Variable: (m, v.size) array}
        self.const = np.asarray(const, dtype=np.float64)
        self.scalar = scalar

    # -- shape -----------------------------------------------------------------------------------------
    @property
    def size(self):
        return len(self.const)

    @property
    def shape(self):
        return () if self.scalar else (self.size,)

    def __len__(self):
        return self.size

    # -- arithmetic ------------------------------------------------------------------------------------
    @staticmethod
    def _lift(x, m):
        if isinstance(x, Expression):
            return x
        c = _const(x)
        if c.ndim != 1:
            raise NotImplementedError("cfmm.cvx: only vectors and scalars")
        return Expression({}, np.broadcast_to(c, (m,)).copy() if len(c) == 1 and m != 1 else c, scalar=(np.ndim(x) == 0))

    def _binary(self, other, sign):
        o = self._lift(other, self.size)
        a, b = self, o
        m = max(a.size, b.size)
        if a.size != b.size:            # scalar broadcast
            if a.size == 1:
                a = Expression({v: np.repeat(c, m, axis=0) for v, c in a.coefs.items()}, np.repeat(a.const, m))
            elif b.size == 1:
                b = Expression({v: np.repeat(c, m, axis=0) for v, c in b.coefs.items()}, np.repeat(b.const, m))
            else:
                raise ValueError(f"cfmm.cvx: shapes {a.size} and {b.size} do not match")
        coefs = {v: c.copy() for v, c in a.coefs.items()}
        for v, c in b.coefs.items():
            coefs[v] = coefs[v] + sign * c if v in coefs else sign * c
        return Expression(coefs, a.const + sign * b.const, scalar=a.scalar and b.scalar)

    def __add__(self, other): return NotImplemented if isinstance(other, (_ScaledAtom, _Concave)) else self._binary(other, 1.0)
    __radd__ = __add__
    def __sub__(self, other): return NotImplemented if isinstance(other, (_ScaledAtom, _Concave)) else self._binary(other, -1.0)
    def __rsub__(self, other): return (-self)._binary(other, 1.0)
    def __neg__(self): return Expression({v: -c for v, c in self.coefs.items()}, -self.const, self.scalar)

    def __mul__(self, k):
        if isinstance(k, Expression):
            raise NotImplementedError("cfmm.cvx: products of expressions are not affine")
        k = np.asarray(k, dtype=np.float64)
        if k.ndim == 0:
            return Expression({v: float(k) * c for v, c in self.coefs.items()}, float(k) * self.const, self.scalar)
        if k.shape != (self.size,):
            raise ValueError("cfmm.cvx: elementwise factor of the wrong length")
        return Expression({v: k[:, None] * c for v, c in self.coefs.items()}, k * self.const)
    __rmul__ = __mul__

    def __truediv__(self, k): return self * (1.0 / np.asarray(k, dtype=np.float64))

    def __rmatmul__(self, M):
        M = np.asarray(M, dtype=np.float64)
        if M.ndim == 1:
            if len(M) != self.size:
                raise ValueError("cfmm.cvx: inner dimensions do not match")
            return Expression({v: (M @ c)[None, :] for v, c in self.coefs.items()}, np.array([M @ self.const]), scalar=True)
        if M.ndim != 2 or M.shape[1] != self.size:
            raise ValueError("cfmm.cvx: inner dimensions do not match")
        return Expression({v: M @ c for v, c in self.coefs.items()}, M @ self.const)

    def __matmul__(self, M):
        M = np.asarray(M, dtype=np.float64)
        if M.ndim != 1:
            raise NotImplementedError("cfmm.cvx: expression @ matrix")
        return self.__rmatmul__(M)

    def __getitem__(self, k):
        if isinstance(k, (int, np.integer)):
            k = int(k) % self.size
            return Expression({v: c[k:k + 1] for v, c in self.coefs.items()}, self.const[k:k + 1], scalar=True)
        return Expression({v: c[k] for v, c in self.coefs.items()}, self.const[k])

    # -- comparisons -> constraints ---------------------------------------------------------------------
    def __ge__(self, other): return Constraint(self - other, ">=")
    def __le__(self, other): return Constraint(self._lift(other, self.size) - self, ">=")
    def __eq__(self, other): return Constraint(self - other, "==")     # noqa: PLW1641 (expressions are not hashed)
    __hash__ = None

    # -- evaluation -------------------------------------------------------------------------------------
    @property
    def value(self):
        out = self.const.copy()
        for v, c in self.coefs.items():
            if v._value is None:
                return None
            out = out + c @ v._value
        return float(out[0]) if self.scalar else out


class Variable(Expression):
    """cp.Variable(n, nonneg=True)   (arbitrage.py:51-52: the tenders Delta_i, Lambda_i)"""
    _count = 0

    def __init__(self, shape=1, nonneg=False, name=None):
        n = int(shape if not isinstance(shape, tuple) else shape[0])
        self.n = n
        self.nonneg = bool(nonneg)
        self._value = None
        Variable._count += 1
        self.name = name or f"var{Variable._count}"
        Expression.__init__(self, {self: np.eye(n)}, np.zeros(n))

    __hash__ = object.__hash__

    def __eq__(self, other):            # identity for dict keys; `var == x` constraints go through Expression
        if isinstance(other, Variable):
            return self is other
        return Expression.__eq__(self, other)

    @property
    def value(self):
        return None if self._value is None else self._value.copy()


class Constraint:
    def __init__(self, expr, op):
        self.expr, self.op = expr, op


class _GeoMean:
    """cp.geo_mean(x, p=w) of an affine x: only ever compared with a constant (arbitrage.py:65,68-70)"""

    def __init__(self, expr, w):
        self.expr, self.w = expr, w

    def __ge__(self, rhs):
        if isinstance(rhs, _GeoMean):
            raise NotImplementedError("cfmm.cvx: geo_mean(x) >= geo_mean(y) needs a constant y")
        return _GeoConstraint(self.expr, self.w, float(rhs))


class _GeoConstraint:
    def __init__(self, expr, w, rhs):
        self.expr, self.w, self.rhs = expr, w, rhs


class _ScaledAtom:
    """k * inv_prod(x) of an affine x: only ever subtracted from sum(x) (the stableswap trading function)"""

    def __init__(self, expr, k=1.0):
        self.expr, self.k = expr, float(k)

    def __mul__(self, k): return _ScaledAtom(self.expr, self.k * float(k))
    __rmul__ = __mul__
    def __neg__(self): return _ScaledAtom(self.expr, -self.k)

    def __rsub__(self, lin):           # lin - k * inv_prod(x)
        if not isinstance(lin, Expression) or lin.size != 1 or not self.k > 0:
            raise NotImplementedError("cfmm.cvx: inv_prod appears only as  sum(x) - alpha * inv_prod(x)  with alpha > 0")
        return _FnExpr("curve", self.expr, self.k, lin)

    def __radd__(self, lin):           # lin + (-k) * inv_prod(x)
        return (-self).__rsub__(lin)


class _Power:
    """cp.power(x, q), elementwise, of an affine x: only ever summed (the power-sum trading function)"""

    def __init__(self, expr, q):
        self.expr, self.q = expr, float(q)


class _FnExpr:
    """a two-asset trading function of x = R + gamma*D - L, waiting for its `>= constant`"""

    def __init__(self, kind, expr, param, lin=None):
        self.kind, self.expr, self.param, self.lin = kind, expr, param, lin

    def __ge__(self, rhs):
        if not np.isscalar(rhs) and not (isinstance(rhs, np.ndarray) and rhs.ndim == 0):
            raise NotImplementedError("cfmm.cvx: a trading function is compared with its value at the current reserves (a number)")
        return _FnConstraint(self.kind, self.expr, self.param, self.lin, float(rhs))


class _FnConstraint:
    def __init__(self, kind, expr, param, lin, rhs):
        self.kind, self.expr, self.param, self.lin, self.rhs = kind, expr, param, lin, rhs


def inv_prod(x):
    """cp.inv_prod(x) = 1 / prod(x): a number for a constant x (the right-hand sides)"""
    if isinstance(x, Expression):
        return _ScaledAtom(x, 1.0)
    return float(1.0 / np.prod(np.asarray(x, dtype=np.float64)))


def power(x, p):
    """cp.power(x, p), elementwise: an array for a constant x"""
    if isinstance(x, Expression):
        if not 0.0 < float(p) < 1.0:
            raise NotImplementedError("cfmm.cvx: power(x, p) of a variable needs 0 < p < 1 (the concave range)")
        return _Power(x, p)
    return np.power(np.asarray(x, dtype=np.float64), float(p))


def geo_mean(x, p=None):
    """cp.geo_mean(x, p): prod_k x_k^(p_k / sum p); a plain number for a constant x (the right-hand sides)"""
    if isinstance(x, Expression):
        n = x.size
        w = np.ones(n) if p is None else np.asarray(p, dtype=np.float64)
        if w.shape != (n,) or not np.all(w > 0):
            raise ValueError("cfmm.cvx: geo_mean weights must be positive, one per entry")
        return _GeoMean(x, w / w.sum())
    a = np.asarray(x, dtype=np.float64)
    w = np.ones(len(a)) if p is None else np.asarray(p, dtype=np.float64)
    return float(np.exp((w / w.sum()) @ np.log(a)))


class _Concave:
    """lin + sum over terms of  sum_j k_j f(x_j),  f = log | square, x an affine vector, lin a scalar affine expression: the
    objective of a separable concave utility.  A term is elementwise (`vector`) until cp.sum / `a @` closes it."""
    __array_ufunc__ = None
    __array_priority__ = 1000

    def __init__(self, terms, lin=None, vector=False):
        self.terms = terms           # [(kind, Expression x, k array of x.size)]
        self.lin = lin               # scalar Expression or None
        self.vector = vector         # one open elementwise term: multiply / sum / @ still apply to it

    def _closed(self, what):
        if self.vector:
            raise ValueError(f"cfmm.cvx: {what} of an elementwise log / square term: close it with cp.sum(...) or `a @ ...` first")

    def _scaled(self, k):
        k = np.asarray(k, dtype=np.float64)
        if k.ndim > 1 or (k.ndim == 1 and not self.vector) or (k.ndim == 1 and k.shape != self.terms[0][2].shape):
            raise ValueError("cfmm.cvx: factor of the wrong shape")
        if self.lin is not None and k.ndim != 0:       # (a linear part exists on closed, scalar terms only: it takes scalar factors)
            raise ValueError("cfmm.cvx: a vector factor on an objective that already holds a linear term")
        return _Concave([(kind, x, kk * k) for kind, x, kk in self.terms], None if self.lin is None else self.lin * float(k), self.vector)

    def __mul__(self, k): return self._scaled(k)
    __rmul__ = __mul__
    def __neg__(self): return self._scaled(-1.0)
    def __truediv__(self, k): return self._scaled(1.0 / float(k))

    def __rmatmul__(self, a):          # a @ log(x)
        return sum(self._scaled(np.asarray(a, dtype=np.float64)))

    def _plus(self, other, sign):
        self._closed("a sum")
        if isinstance(other, _Concave):
            other._closed("a sum")
            lin = other.lin * sign if other.lin is not None else None
            if self.lin is not None:
                lin = self.lin if lin is None else self.lin + lin
            return _Concave(self.terms + [(kind, x, sign * k) for kind, x, k in other.terms], lin)
        o = Expression._lift(other, 1)
        if o.size != 1:
            raise ValueError("cfmm.cvx: the objective must be a scalar")
        return _Concave(list(self.terms), o * sign if self.lin is None else self.lin + o * sign)

    def __add__(self, other): return self._plus(other, 1.0)
    __radd__ = __add__
    def __sub__(self, other): return self._plus(other, -1.0)
    def __rsub__(self, other): return (-self)._plus(other, 1.0)

    @property
    def size(self):
        return self.terms[0][1].size if self.vector else 1

    @property
    def value(self):
        out = 0.0 if self.lin is None else self.lin.value
        for kind, x, k in self.terms:
            xv = x.value
            if xv is None or out is None:
                return None
            f = k * (np.log(xv) if kind == "log" else np.square(xv))
            if self.vector:
                return f
            out = out + float(np.sum(f))
        return out


def log(x):
    """cp.log, elementwise -- as an objective term of entries of psi (the utility table's CFMM_ULOG)"""
    if not isinstance(x, Expression):
        return np.log(x)
    return _Concave([("log", x, np.ones(x.size))], vector=True)


def square(x):
    """cp.square, elementwise -- as a (negatively weighted) objective term of entries of psi (CFMM_UQUAD)"""
    if not isinstance(x, Expression):
        return np.square(x)
    return _Concave([("square", x, np.ones(x.size))], vector=True)


def sum_squares(x):
    return sum(square(x)) if isinstance(x, Expression) else float(np.sum(np.square(x)))


def multiply(a, x):
    """cp.multiply: elementwise product with a constant"""
    if isinstance(a, (Expression, _Concave)) and not isinstance(x, (Expression, _Concave)):
        a, x = x, a
    if isinstance(a, (Expression, _Concave)):
        raise NotImplementedError("cfmm.cvx: products of expressions")
    return x * a if isinstance(x, (Expression, _Concave)) else np.multiply(a, x)


def sum(x, axis=None):        # noqa: A001 (mirrors cp.sum)
    """cp.sum: a list of expressions adds elementwise (arbitrage.py:54); an expression or array sums its entries"""
    if isinstance(x, (list, tuple)):
        return builtins.sum(x[1:], x[0])
    if isinstance(x, _Power):
        return _FnExpr("powersum", x.expr, 1.0 - x.q)
    if isinstance(x, _Concave):
        return _Concave(list(x.terms), x.lin) if x.vector else x
    if isinstance(x, Expression):
        return np.ones(x.size) @ x
    return float(np.sum(x))


class Maximize:
    def __init__(self, expr):
        self.terms = []
        self.whole = expr
        if isinstance(expr, _Concave):
            expr._closed("an objective")
            for kind, x, k in expr.terms:
                if (kind == "log" and np.any(k < 0)) or (kind == "square" and np.any(k > 0)):
                    raise ValueError("cfmm.cvx: the objective is not concave (log terms take weights >= 0, squares <= 0)")
            self.terms = [(kind, x, k) for kind, x, k in expr.terms]
            expr = expr.lin if expr.lin is not None else Expression({}, np.zeros(1), scalar=True)
        if not isinstance(expr, Expression) or expr.size != 1:
            raise ValueError("cfmm.cvx: the objective must be a scalar affine expression (plus log / square terms of entries of psi)")
        self.expr = expr             # the linear part
        self.sign = 1.0

    @property
    def value(self):
        return self.whole.value


class Minimize(Maximize):
    def __init__(self, expr):
        Maximize.__init__(self, -expr if isinstance(expr, (Expression, _Concave)) else expr)
        self.sign = -1.0

    @property
    def value(self):
        v = self.whole.value
        return None if v is None else -v


def _is_scaled_identity(c, tol=1e-12):
    """c == g * I ?  -> g or None"""
    if c.shape[0] != c.shape[1]:
        return None
    g = c[0, 0]
    return float(g) if np.abs(c - g * np.eye(c.shape[0])).max() <= tol * max(1.0, abs(g)) else None


def _pool_of(expr):
    """expr == R + gamma * Delta - Lambda  ->  (Delta, Lambda, R, gamma)   (arbitrage.py:60)"""
    if len(expr.coefs) != 2:
        return None
    (va, ca), (vb, cb) = expr.coefs.items()
    ga, gb = _is_scaled_identity(ca), _is_scaled_identity(cb)
    if ga is None or gb is None:
        return None
    if abs(gb + 1.0) <= 1e-12 and 0.0 < ga <= 1.0 + 1e-12:
        D, L, g = va, vb, ga
    elif abs(ga + 1.0) <= 1e-12 and 0.0 < gb <= 1.0 + 1e-12:
        D, L, g = vb, va, gb
    else:
        return None
    if not (isinstance(D, Variable) and isinstance(L, Variable) and D.nonneg and L.nonneg and np.all(expr.const > 0)):
        return None
    return D, L, expr.const.copy(), min(g, 1.0)


class Problem:
    """cp.Problem(obj, cons); .solve() -> prob.value (arbitrage.py:81-84)"""

    def __init__(self, objective, constraints=()):
        self.objective = objective
        self.constraints = list(constraints)
        self.value = None
        self.status = None
        self.routing = None          # the cfmm.Problem the model was mapped onto (prices: .routing.nu)

    # -- the pattern match --------------------------------------------------------------------------------
    def _match(self):
        pools = {}                   # Delta variable -> dict
        rest = []
        vec_nonneg = []
        for con in self.constraints:
            if isinstance(con, _GeoConstraint):
                pl = _pool_of(con.expr)
                if pl is None:
                    raise NotImplementedError("cfmm.cvx: geo_mean(...) must be taken of R + gamma*Delta - Lambda (arbitrage.py:60,65)")
                D, L, R, g = pl
                want = float(np.exp(con.w @ np.log(R)))
                if abs(con.rhs - want) > 1e-9 * want:
                    raise NotImplementedError("cfmm.cvx: the right-hand side must be the pool's trading function at its current "
                                              f"reserves ({want:.12g}), got {con.rhs:.12g}")
                if D in pools:
                    raise NotImplementedError("cfmm.cvx: two trading functions for one pool")
                pools[D] = dict(D=D, L=L, R=R, fee=g, kind="geomean", w=con.w)
            elif isinstance(con, _FnConstraint):
                pl = _pool_of(con.expr)
                # (power sum: two assets; stableswap: 2..8 -- three and more ride in the K-asset table, csrc/phik.hpp)
                if pl is None or (con.expr.size != 2 and not (con.kind == "curve" and 3 <= con.expr.size <= 8)):
                    raise NotImplementedError("cfmm.cvx: a stableswap function must be taken of R + gamma*Delta - Lambda over 2..8 tokens, a power-sum function over two")
                D, L, R, g = pl
                if con.kind == "curve":
                    lin = con.lin                  # must be sum(x) of the SAME x
                    same = set(lin.coefs) == set(con.expr.coefs) and abs(lin.const[0] - R.sum()) <= 1e-12 * R.sum() and \
                        all(np.allclose(lin.coefs[v], np.ones(con.expr.size) @ con.expr.coefs[v]) for v in lin.coefs)
                    if not same:
                        raise NotImplementedError("cfmm.cvx: inv_prod(x) must be subtracted from sum(x) of the same x")
                    want = float(R.sum() - con.param / np.prod(R))
                else:
                    want = float(np.sum(R ** (1.0 - con.param)))
                if abs(con.rhs - want) > 1e-9 * max(1.0, abs(want)):
                    raise NotImplementedError("cfmm.cvx: the right-hand side must be the pool's trading function at its current "
                                              f"reserves ({want:.12g}), got {con.rhs:.12g}")
                if D in pools:
                    raise NotImplementedError("cfmm.cvx: two trading functions for one pool")
                pools[D] = dict(D=D, L=L, R=R, fee=g, kind=con.kind, w=None, param=float(con.param))
            elif isinstance(con, Constraint):
                pl = _pool_of(con.expr) if (con.op == ">=" and con.expr.size > 1) else None
                if pl is not None:
                    vec_nonneg.append(pl)          # new_reserves >= 0: the partner of a constant-sum constraint
                else:
                    rest.append(con)
            else:
                raise NotImplementedError(f"cfmm.cvx: unsupported constraint {type(con).__name__}")
        # constant sum: sum(R + gamma*D - L) >= sum(R), recognised through its >= 0 partner (arbitrage.py:73-74)
        for D, L, R, g in vec_nonneg:
            hit = None
            for con in rest:
                e = con.expr
                if con.op == ">=" and e.size == 1 and set(e.coefs) == {D, L} and abs(e.const[0]) <= 1e-12 * R.sum() \
                        and np.allclose(e.coefs[D], g) and np.allclose(e.coefs[L], -1.0):
                    hit = con
                    break
            if hit is None:
                raise NotImplementedError("cfmm.cvx: `R + gamma*D - L >= 0` without its `sum(...) >= sum(R)` constraint")
            rest.remove(hit)
            if D in pools:
                raise NotImplementedError("cfmm.cvx: two trading functions for one pool")
            if not 2 <= len(R) <= 8:
                raise NotImplementedError("cfmm.cvx: constant-sum pools hold 2..8 tokens")
            pools[D] = dict(D=D, L=L, R=R, fee=g, kind="sum", w=None)
        pools = list(pools.values())
        if not pools:
            raise NotImplementedError("cfmm.cvx: no pool constraints found")
        Lam = {p["L"]: i for i, p in enumerate(pools)}
        Del = {p["D"]: i for i, p in enumerate(pools)}
        width = builtins.sum(len(p["R"]) for p in pools)
        offs = np.cumsum([0] + [len(p["R"]) for p in pools])

        def row_matrix(expr):
            """rows of expr as coefficients on the stacked net-trade slots (Lambda_i - Delta_i): the coefficient on Lambda_i,
            which the one on Delta_i must negate (arbitrage.py:54: only A_i (Lambda_i - Delta_i) leaves a pool)"""
            M = np.zeros((expr.size, width))
            for v, c in expr.coefs.items():
                if v in Lam:
                    i, other = Lam[v], pools[Lam[v]]["D"]
                    M[:, offs[i]:offs[i + 1]] = c
                elif v in Del:
                    i, other = Del[v], pools[Del[v]]["L"]
                else:
                    raise NotImplementedError("cfmm.cvx: a variable that is no pool's tender")
                co = expr.coefs.get(other)
                if co is None or np.abs(co + c).max() > 1e-12:
                    raise NotImplementedError("cfmm.cvx: only the net trade Lambda - Delta may appear outside the pool constraints")
            return M

        # token rows: every row of the remaining constraints must select entries of psi (0/1 on the slots)
        rows, hs, kinds = [], [], []
        for con in rest:
            M = row_matrix(con.expr)
            for r in range(con.expr.size):
                rows.append(M[r]); hs.append(con.expr.const[r]); kinds.append(_GE if con.op == ">=" else _EQ)
        objM = row_matrix(self.objective.expr)[0]
        tokens = []                  # list of (selector row, h, ctype)
        for r, h, k in zip(rows, hs, kinds):
            if not np.all((np.abs(r) < 1e-12) | (np.abs(r - 1.0) < 1e-12)):
                raise NotImplementedError("cfmm.cvx: constraints on psi must be on its entries")
            if any(np.abs(r - t[0]).max() < 1e-12 for t in tokens):
                raise NotImplementedError("cfmm.cvx: two constraints on one entry of psi")
            tokens.append((np.round(r), float(h), k))
        # separable concave terms of the objective (the utility table): every row of a term's argument is ONE entry of psi, plus a
        # constant for the logarithm (psi_j + h_j) and none for the square; such an entry takes no other constraint -- the
        # logarithm's domain is its constraint, the square's entry is free
        table = {}                   # token index -> weight k_j
        for kind, x, k in self.objective.terms:
            M = row_matrix(x)
            for r in range(x.size):
                if k[r] == 0.0:
                    continue
                row, kr = M[r], float(k[r])
                nz = row[np.abs(row) > 1e-12]
                if kind == "square" and len(nz) and np.all(np.abs(nz - nz[0]) <= 1e-12 * abs(nz[0])):
                    row, kr = row / nz[0], kr * nz[0] ** 2       # square(s psi_j) = s^2 psi_j^2
                if not np.all((np.abs(row) < 1e-12) | (np.abs(row - 1.0) < 1e-12)) or not np.any(row > 0.5):
                    raise NotImplementedError(f"cfmm.cvx: cp.{kind}(...) must be taken of entries of psi")
                if any(np.abs(row - t[0]).max() < 1e-12 for t in tokens):
                    raise NotImplementedError(f"cfmm.cvx: an entry of psi under cp.{kind}(...) takes no other constraint or term")
                if kind == "log":
                    if x.const[r] < 0.0:
                        raise NotImplementedError("cfmm.cvx: log(psi_j + h_j) needs h_j >= 0")
                    tokens.append((np.round(row), float(x.const[r]), _ULOG))
                else:
                    if x.const[r] != 0.0:
                        raise NotImplementedError("cfmm.cvx: square(...) must be taken of psi_j itself")
                    tokens.append((np.round(row), 1.0 / (2.0 * -kr), _UQUAD))      # - k psi^2 = - psi^2 / (2 depth)
                table[len(tokens) - 1] = (kind, kr)
        # the objective: a combination of token rows, plus -- at most -- selector rows of tokens no constraint mentions
        left = objM.copy()
        c = np.zeros(len(tokens))
        if tokens:
            T = np.stack([t[0] for t in tokens], axis=1)
            c, *_ = np.linalg.lstsq(T, objM, rcond=None)
            c[np.abs(c) < 1e-14] = 0.0
            left = objM - T @ c
        c = list(c)
        if np.abs(left).max() > 1e-10:
            # what is left must itself be selector rows times one value each: split by slot ownership
            covered = np.zeros(width, dtype=bool)
            for t in tokens:
                covered |= t[0] > 0.5
            free_slots = np.flatnonzero(~covered & (np.abs(left) > 1e-12))
            vals = np.unique(np.round(left[free_slots], 12))
            for val in vals:
                sel = np.zeros(width); sel[free_slots[np.abs(left[free_slots] - val) < 1e-12]] = 1.0
                tokens.append((sel, 0.0, _FREE)); c.append(float(val))
            left = objM - np.stack([t[0] for t in tokens], axis=1) @ np.asarray(c)
            if np.abs(left).max() > 1e-10:
                raise NotImplementedError("cfmm.cvx: the objective must be linear in psi")
        c = np.asarray(c) * 1.0
        for j, (kind, k) in table.items():
            if kind == "log":        # u = k log(psi + h): the entry's c is the weight; a linear term on the same entry is another utility
                if abs(c[j]) > 1e-12:
                    raise NotImplementedError("cfmm.cvx: a linear term on an entry of psi that sits under cp.log(...)")
                c[j] = k
        if np.any(c < -1e-14):
            raise NotImplementedError("cfmm.cvx: negative objective weights on psi (the utility table's entries take a marginal value c >= 0 as "
                                      "well: CFMM_UQUAD's linear coefficient, CFMM_ULOG's weight)")
        # slots no token row covers: a pool asset that appears in neither objective nor constraints is unpriced
        cover = np.zeros(width)
        for t in tokens:
            cover += t[0]
        if np.any(cover < 0.5):
            raise NotImplementedError("cfmm.cvx: a pool asset that no entry of psi mentions (unbounded or irrelevant)")
        if np.any(cover > 1.5):
            raise NotImplementedError("cfmm.cvx: a pool slot mapped to two tokens")
        local = []
        for i, p in enumerate(pools):
            idx = []
            for s in range(offs[i], offs[i + 1]):
                idx.append(int(np.flatnonzero([t[0][s] > 0.5 for t in tokens])[0]))
            local.append(idx)
        n = len(tokens)
        util = _Utility(np.maximum(c, 0.0), np.array([t[1] for t in tokens]), np.array([t[2] for t in tokens], dtype=np.int32))
        return pools, local, n, util

    def solve(self, solver=None, verbose=False, **kw):
        """maps the model onto cfmm.Problem and solves it on the device; returns prob.value like cvxpy"""
        pools, local, n, util = self._match()
        tol = float(kw.pop("tol", 1e-9))
        key = _pool_signature(n, local, pools) if RESIDENT_MAX > 0 else None
        p = _resident.get(key) if key is not None else None
        if p is not None and p.ctx is not None:          # the same pools as an earlier model (and not closed since): utility + warm start
            p.set_utility(util)
            kw.setdefault("warm_start", True)
        else:
            p = _RoutingProblem(n, local, [pl["R"] for pl in pools], [pl["fee"] for pl in pools],
                                [pl["kind"] for pl in pools], [pl["w"] for pl in pools],
                                [pl.get("param") for pl in pools], utility=util)
            if CONTEXT_FACTORY is not None:
                p.ctx = CONTEXT_FACTORY(n)
            if key is not None:
                _resident.pop(key, None)
                while len(_resident) >= RESIDENT_MAX:    # (the oldest entry goes; whoever still holds it through prob.routing keeps it alive)
                    _resident.pop(next(iter(_resident)))
                _resident[key] = p
        p.solve(tol=tol, **kw)
        self.routing = p
        deltas, lambdas = p.deltas, p.lambdas
        for pl, d, l in zip(pools, deltas, lambdas):
            pl["D"]._value = np.array(d, dtype=np.float64)       # (copies: the resident problem's arrays belong to its next solve)
            pl["L"]._value = np.array(l, dtype=np.float64)
        self.status = {"optimal": OPTIMAL, "inaccurate": INACCURATE, "infeasible": INFEASIBLE}.get(p.status, p.status)
        self.value = self.objective.value
        if verbose:
            print(f"cfmm.cvx: {len(pools)} pools / {n} tokens, status {p.status}, gap {p.gap:.2e}, infeas {p.infeas:.2e}, "
                  f"{p.stats['evals']} dual evaluations")
        return self.value
