"""TEST INFRASTRUCTURE (oracle side) -- an independent minimisation of the routing program's DUAL by SciPy's L-BFGS-B over the NumPy
restatements of the per-pool subproblems (oracle/pools_np.py).

Only `tests/`, `tools/fuzz_*.py`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

Why it exists (round 6): the fuzz campaigns' referee was the SciPy PRIMAL (oracle/primal_scipy.py, SLSQP on the
program exactly as /root/reference/arbitrage.py:51-78 writes it) -- and SLSQP gives up on two thirds of the small random instances, so
most fuzzed instances were checked only by the product's own two outer iterations against each other.  The dual of the decomposed
program,

    minimise over log-prices s inside the utility's box   g(s) = sum_i arb_i(e^{s[l_i]}) + (e^s - c)'h,     grad = nu (psi + h),

is an unconstrained-but-for-bounds convex problem of n variables that a quasi-Newton method finishes on every instance; by strong
duality its minimum is the program's optimal value.  It shares nothing with the device's outer iterations (its own projected L-BFGS,
its barrier Newton) but the mathematics, and nothing with the device's pool solvers: arb_i are the restatements the C twin and the
kernels are pinned against.  Where pools are piecewise linear (constant sum) g has kinks and the minimiser's PRIMAL point is not
recovered -- the dual VALUE still is (to ~1e-7 relative), which is what a referee of the optimal value needs.

PARITY UNPINNED by the reference (cvxpy is not installed here); pinned against the 50-digit KKT optima of the shipped instances and
against the SciPy primal wherever that succeeds (tests/test_oracle.py).
"""
import numpy as np

from oracle import pools_np as P

GE, EQ, FREE = 0, 1, 2


def pool_eval(kind, R, w, gamma, param, p):
    """one pool of the reference's vocabulary at local prices p -> (y = Lambda - Delta per leg, arb = p'y)"""
    k = len(R)
    if kind in ("geomean", "product"):
        if k == 2:
            ya, yb, arb = P.arb_geomean2(R[0], R[1], gamma, w[0], w[1], p[0], p[1])
            return np.array([float(ya), float(yb)]), float(arb)
        y, arb = P.arb_geomean_n(R, w, gamma, p)
        return np.asarray(y, float), float(arb)
    if kind == "sum":
        return P.arb_sum(R, gamma, p)
    if kind == "curve":
        if k == 2:
            ya, yb, arb = P.arb_curve2(R[0], R[1], gamma, param, p[0], p[1])
            return np.array([float(ya), float(yb)]), float(arb)
        y, arb = P.arb_stable_n(np.asarray(R, float)[:, None], np.array([param]), np.array([gamma]), np.asarray(p, float)[:, None])
        return y[:, 0], float(arb[0])
    if kind == "powersum":
        ya, yb, arb = P.arb_power2(R[0], R[1], gamma, param, p[0], p[1])
        return np.array([float(ya), float(yb)]), float(arb)
    raise ValueError(f"pool kind {kind!r}")


def dual_eval(inst, nu):
    """(sum_i arb_i(nu), psi(nu)) of a normalised instance (oracle/instances.py: normalise; `params` per pool where the kind has one)"""
    n = inst["n_tokens"]
    psi = np.zeros(n)
    f = 0.0
    params = inst.get("params") or [None] * len(inst["local_indices"])
    for l, R, g, kind, w, prm in zip(inst["local_indices"], inst["reserves"], inst["fees"], inst["kinds"], inst["weights"], params):
        y, arb = pool_eval(kind, np.asarray(R, float), np.asarray(w, float), float(g), prm, nu[np.asarray(l)])
        np.add.at(psi, np.asarray(l), y)
        f += arb
    return f, psi


def solve_dual(inst, nu0=None, restarts=3):
    """-> dict(value (the dual optimum = the program's optimal value), nu, psi, gap, infeas, pg, converged, evals).
    `converged`: the projected gradient in log-prices is below 1e-6 of the dual's scale, or SciPy reports convergence of the value."""
    from scipy.optimize import minimize
    n = inst["n_tokens"]
    c, h, ct = np.asarray(inst["c"], float), np.asarray(inst["h"], float), np.asarray(inst["ctype"], int)
    if ((ct == FREE) & ~(c > 0)).any():
        raise ValueError("an unconstrained token with c = 0: unbounded")
    span = 60.0
    cmax = float(c.max()) if (c > 0).any() else 1.0
    mid = np.log(cmax)
    lo = np.where((ct == GE) & (c > 0), np.log(np.where(c > 0, c, 1.0)), mid - span)
    hi = np.full(n, mid + span)
    fixed = ct == FREE
    lo = np.where(fixed, np.log(np.where(c > 0, c, 1.0)), lo); hi = np.where(fixed, lo, hi)
    evals = [0]

    def fg(s):
        evals[0] += 1
        nu = np.exp(s)
        f, psi = dual_eval(inst, nu)
        g = f + float((nu - c) @ h)
        G = nu * (psi + h)
        G[fixed] = 0.0
        return g, G

    s = np.log(np.asarray(nu0, float)) if nu0 is not None else np.where(c > 0, np.log(np.where(c > 0, c, 1.0)), mid)
    s = np.clip(s, lo, hi)
    best = None
    for _ in range(max(1, restarts)):
        r = minimize(fg, s, jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)), options=dict(maxiter=4000, maxfun=20000, ftol=1e-16, gtol=1e-12, maxcor=20))
        s = r.x
        if best is None or r.fun < best[0]:
            best = (float(r.fun), r.x.copy(), bool(r.success))
    g, s, ok = best
    nu = np.exp(s)
    f, psi = dual_eval(inst, nu)
    G = nu * (psi + h); G[fixed] = 0.0
    pg = np.where(s <= lo + 1e-12, np.minimum(G, 0.0), np.where(s >= hi - 1e-12, np.maximum(G, 0.0), G))
    scale = max(1.0, abs(g))
    r_ = psi + h
    cs = float((nu - c) @ r_)
    viol = float(np.where(ct == GE, np.maximum(-r_, 0.0), np.where(ct == EQ, np.abs(r_), 0.0)).max())
    den = max(float(np.abs(psi).max()), float(np.abs(h).max()), 1e-300)
    return dict(value=g, nu=nu, psi=psi, gap=abs(cs) / scale, infeas=viol / den, pg=float(np.abs(pg).max()) / scale,
                converged=bool(ok or float(np.abs(pg).max()) <= 1e-6 * scale), evals=evals[0])
