"""TEST INFRASTRUCTURE -- CPU oracle, part 2: the reference's PRIMAL model through SciPy.

PARITY UNPINNED (see oracle/pools_np.py).  This is the closest thing to "running the
reference" this container allows: the optimisation model below is a statement-by-statement
restatement of /root/reference/arbitrage.py:51-78 (variables :51-52, psi :54, objective :57,
new reserves :60, trading-function constraints :63-74, utility constraints :77), of
liquidation.py:57,77-80 and of two-asset.py:66,74,86 -- handed to SciPy's SLSQP instead of
cvxpy/ECOS.  It shares no code and no algorithm with the dual-decomposition path
(oracle/pools_np.py, oracle/cfmm_oracle.c, the HIP kernels), which is what makes the
agreement of the two a meaningful check.  Practical up to ~50 pools.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may use oracle/.
"""
import numpy as np
from scipy.optimize import minimize


def solve_primal(inst, x0=None, ftol=1e-15, maxiter=2000):
    """inst: a dict produced by oracle.instances.normalise (or the same fields).
    Extra pool kinds: "curve" takes inst["params"][i] = alpha (x + y - alpha / (xy)); "powersum" takes params[i] = t
    (x^(1-t) + y^(1-t)).
    Returns dict(value, psi, deltas, lambdas, y) with y[i] = lambdas[i] - deltas[i]."""
    n = inst["n_tokens"]
    L = inst["local_indices"]
    R = inst["reserves"]
    G = np.asarray(inst["fees"], float)
    K = inst["kinds"]
    W = inst["weights"]
    P = inst.get("params", [None] * len(L))
    c, h, ctype = inst["c"], inst["h"], inst["ctype"]
    m = len(L)
    sizes = [len(l) for l in L]
    off = np.concatenate([[0], np.cumsum(sizes)])
    nnz = int(off[-1])
    gidx = np.concatenate(L)
    gam = np.concatenate([np.full(s, g) for s, g in zip(sizes, G)])
    Rf = np.concatenate(R)

    # z = [D (nnz) | Lam (nnz)]
    def psi_of(z):
        y = z[nnz:] - z[:nnz]
        return np.bincount(gidx, weights=y, minlength=n)

    # utility: c'psi over the box the constraints below state -- the reference's (arbitrage.py:57,77) -- plus, per token, an entry
    # of the separable table (NOT in the reference: ctype 3: c log(psi + h); 4: c psi - psi^2 / (2 h))
    lin = ctype <= 2
    lg, qd = ctype == 3, ctype == 4

    def u_and_grad(psi):
        val = float(c[lin] @ psi[lin])
        g = np.where(lin, c, 0.0)
        if lg.any():
            arg = np.maximum(psi[lg] + h[lg], 1e-300)
            val += float(c[lg] @ np.log(arg)); g[lg] = c[lg] / arg
        if qd.any():
            val += float(c[qd] @ psi[qd] - 0.5 * psi[qd] ** 2 @ (1.0 / h[qd])); g[qd] = c[qd] - psi[qd] / h[qd]
        return val, g

    def fobj(z):
        return -u_and_grad(psi_of(z))[0]

    def gobj(z):
        g = u_and_grad(psi_of(z))[1][gidx]
        return -np.concatenate([-g, g])          # d U(psi) / dz,  psi = sum A (Lam - D)

    def newres(z):
        return Rf + gam * z[:nnz] - z[nnz:]

    TINY = 1e-300

    def cons_f(z):
        x = newres(z)
        out = []
        for i in range(m):
            s = slice(off[i], off[i + 1])
            xi, Ri = x[s], R[i]
            if K[i] == "geomean":
                out.append(np.sum(W[i] * (np.log(np.maximum(xi, TINY)) - np.log(Ri))))
            elif K[i] == "sum":
                out.append(np.sum(xi) - np.sum(Ri))
                out.extend(xi)                                   # new reserves >= 0
            elif K[i] == "curve":
                al = P[i]
                xp = np.maximum(xi, 1e-9 * Ri)                   # inv_prod's domain is x > 0
                out.append(np.sum(xi) - al / np.prod(xp) - (np.sum(Ri) - al / np.prod(Ri)))
                out.extend(xi - 1e-9 * Ri)
            elif K[i] == "powersum":
                q = 1.0 - P[i]
                xp = np.maximum(xi, 1e-12 * Ri)
                out.append((np.sum(xp ** q) - np.sum(Ri ** q)) / np.sum(Ri ** q))
                out.extend(xi - 1e-12 * Ri)
            else:
                raise ValueError(K[i])
        psi = psi_of(z)
        out.extend((psi + h)[ctype == 0])
        out.extend((psi + h)[lg] - 1e-12 * (1.0 + np.abs(h[lg])))        # log's domain
        return np.asarray(out, float)

    def cons_j(z):
        x = newres(z)
        rows = []
        for i in range(m):
            s = slice(off[i], off[i + 1])
            xi = x[s]

            def row_from(dphi):
                r = np.zeros(2 * nnz)
                r[s] = dphi * G[i]
                r[nnz + off[i]: nnz + off[i + 1]] = -dphi
                return r
            if K[i] == "geomean":
                rows.append(row_from(W[i] / np.maximum(xi, TINY)))
            elif K[i] == "sum":
                rows.append(row_from(np.ones(sizes[i])))
                for k in range(sizes[i]):
                    e = np.zeros(sizes[i]); e[k] = 1.0
                    rows.append(row_from(e))
            elif K[i] == "curve":
                al = P[i]
                xp = np.maximum(xi, 1e-9 * R[i])
                rows.append(row_from(1.0 + al / (np.prod(xp) * xp)))
                for k in range(sizes[i]):
                    e = np.zeros(sizes[i]); e[k] = 1.0
                    rows.append(row_from(e))
            elif K[i] == "powersum":
                q = 1.0 - P[i]
                xp = np.maximum(xi, 1e-12 * R[i])
                rows.append(row_from(q * xp ** (q - 1.0) / np.sum(R[i] ** q)))
                for k in range(sizes[i]):
                    e = np.zeros(sizes[i]); e[k] = 1.0
                    rows.append(row_from(e))
        for k in list(np.where(ctype == 0)[0]) + list(np.where(lg)[0]):
            r = np.zeros(2 * nnz)
            sel = gidx == k
            r[:nnz][sel] = -1.0
            r[nnz:][sel] = 1.0
            rows.append(r)
        return np.asarray(rows)

    eq_tokens = np.where(ctype == 1)[0]

    def eq_f(z):
        return (psi_of(z) + h)[eq_tokens]

    def eq_j(z):
        rows = []
        for k in eq_tokens:
            r = np.zeros(2 * nnz)
            sel = gidx == k
            r[:nnz][sel] = -1.0
            r[nnz:][sel] = 1.0
            rows.append(r)
        return np.asarray(rows)

    cons = [dict(type="ineq", fun=cons_f, jac=cons_j)]
    if len(eq_tokens):
        cons.append(dict(type="eq", fun=eq_f, jac=eq_j))
    if x0 is None:
        x0 = np.zeros(2 * nnz)
    res = minimize(fobj, x0, jac=gobj, bounds=[(0, None)] * (2 * nnz), constraints=cons,
                   method="SLSQP", options=dict(ftol=ftol, maxiter=maxiter))
    z = res.x
    D = [z[off[i]:off[i + 1]].copy() for i in range(m)]
    Lam = [z[nnz + off[i]: nnz + off[i + 1]].copy() for i in range(m)]
    return dict(value=-res.fun, psi=psi_of(z), deltas=D, lambdas=Lam,
                y=[l - d for l, d in zip(Lam, D)], success=bool(res.success),
                message=res.message, nit=res.nit)
