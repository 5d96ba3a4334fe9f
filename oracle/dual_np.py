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


def arb_stable_1(R, alpha, gamma, p):
    """ONE K-asset stableswap pool (phi = sum x - alpha / prod x) at local prices p: the two scalar equations of oracle/pools_np.py:
    arb_stable_n -- multiplier m and coupling s = alpha / prod x, x_j = clip(R_j, s / (p_j / (gamma m) - 1), s / (p_j / m - 1)) -- solved
    with SciPy's brentq inside a bisection, in plain Python floats, instead of 80 x 64 vectorised bisection steps: the same root to
    rounding (pinned against arb_stable_n in tests/test_oracle.py), ~50x faster for one pool, which is what a referee called thousands
    of times needs"""
    import math
    from scipy.optimize import brentq
    R = [float(x) for x in R]; p = [float(x) for x in p]
    k = len(R)
    lR = [math.log(x) for x in R]; la = math.log(alpha)
    lsR = la - sum(lR); sR = math.exp(lsR)
    lmax = math.log(min(p) / gamma)
    NINF = float("-inf")

    def legs(lm, ls):
        em = math.exp(-lm)
        lx, op = [0.0] * k, False
        for j in range(k):
            qw = p[j] * em; qd = qw / gamma
            gd = math.log(max(qd - 1.0, 1e-300)) if qd > 1.0 else NINF
            gw = math.log(max(qw - 1.0, 1e-300)) if qw > 1.0 else NINF
            if gd == NINF:
                op = True
            lx[j] = min(max(lR[j], ls - gd), ls - gw)          # (np.clip's order: the lower bound first, then the upper)
        return lx, op

    def gap(lm):
        h = lambda ls: ls - (la - sum(legs(lm, ls)[0]))
        a, b = lsR - 90.0, lsR + 90.0
        ha, hb = h(a), h(b)
        ls = brentq(h, a, b, xtol=1e-15, rtol=4 * 2.220446049250313e-16, maxiter=200) if ha < 0.0 < hb else (a if ha >= 0.0 else b)
        lx, op = legs(lm, ls)
        if op:
            return 1.0, lx
        dx = sum((math.exp(min(lx[j], 700.0)) - R[j]) for j in range(k) if lx[j] != lR[j])
        return dx - (math.exp(ls) - sR), lx

    lo, hi = lmax - 90.0, lmax
    for _ in range(64):
        lm = 0.5 * (lo + hi)
        if gap(lm)[0] < 0.0:
            lo = lm
        else:
            hi = lm
    lx = gap(hi)[1]
    y = np.zeros(k)
    for j in range(k):
        d = 0.0 if lx[j] == lR[j] else R[j] - math.exp(lx[j])
        y[j] = d if d > 0.0 else d / gamma
    return y, float(np.dot(p, y))


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
            y, arb = P.arb_curve2(R[0], R[1], gamma, param, p[0], p[1])
            return np.asarray(y, float), float(arb)
        return arb_stable_1(R, param, gamma, p)
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
    free_idx = np.flatnonzero(hi > lo)
    G0 = fg(s)[1]
    pg0 = np.where(s <= lo + 1e-12, np.minimum(G0, 0.0), np.where(s >= hi - 1e-12, np.maximum(G0, 0.0), G0))
    if float(np.abs(pg0).max()) > 1e-6 * max(1.0, abs(g)) and 0 < len(free_idx) <= 12:
        # the quasi-Newton iteration stops short where the minimum sits ON a kink of a piecewise-linear pool (all three shipped scripts):
        # a derivative-free polish over the few free log-prices finishes the VALUE (Powell's direction set: no gradient to be fooled)
        def f_only(z):
            t = s.copy(); t[free_idx] = np.clip(z, lo[free_idx], hi[free_idx])
            return fg(t)[0]
        for _ in range(2):
            r = minimize(f_only, s[free_idx], method="Powell", options=dict(xtol=1e-12, ftol=1e-14, maxiter=4000, maxfev=4000))
            if r.fun < g:
                g = float(r.fun); s = s.copy(); s[free_idx] = np.clip(r.x, lo[free_idx], hi[free_idx])
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
                converged=bool(float(np.abs(pg).max()) <= 1e-6 * scale), lbfgsb_success=bool(ok), evals=evals[0])


def solve_dual_network(net, c, h=None, ctype=None, nu0=None, maxiter=3000):
    """the same referee for a whole synthetic network (cfmm/synthetic.py's SoA buckets of the reference's pool kinds: cp2, w2, gn) through the
    VECTORISED restatement oracle/pools_np.py: dual_eval_network -- fast enough for BASELINE's full sizes (C3: ~0.1 s per evaluation on one
    thread, ~60-150 evaluations).  -> dict(value, nu, psi, gap, infeas, pg, evals, lbfgsb_success).  Independent of the device's outer
    iterations AND of the C twin's (which mirrors the device's update by design): SciPy's L-BFGS-B, its own line search, its own stopping."""
    from scipy.optimize import minimize
    n = net["n_tokens"]
    c = np.asarray(c, float)
    h = np.zeros(n) if h is None else np.asarray(h, float)
    ct = np.zeros(n, int) if ctype is None else np.asarray(ctype, int)
    span = 60.0
    mid = np.log(float(c.max())) if (c > 0).any() else 0.0
    lo = np.where((ct == GE) & (c > 0), np.log(np.where(c > 0, c, 1.0)), mid - span)
    hi = np.full(n, mid + span)
    fixed = ct == FREE
    lo = np.where(fixed, np.log(np.where(c > 0, c, 1.0)), lo); hi = np.where(fixed, lo, hi)
    evals = [0]

    def fg(s):
        evals[0] += 1
        nu = np.exp(s)
        psi, f, _ = P.dual_eval_network(net, nu)
        G = nu * (psi + h)
        G[fixed] = 0.0
        return f + float((nu - c) @ h), G

    s = np.clip(np.log(np.asarray(nu0, float)) if nu0 is not None else np.where(c > 0, np.log(np.where(c > 0, c, 1.0)), mid), lo, hi)
    best = None
    for _ in range(2):
        r = minimize(fg, s, jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)), options=dict(maxiter=maxiter, maxfun=4 * maxiter, ftol=1e-15, gtol=1e-10, maxcor=10))
        s = r.x
        if best is None or r.fun < best[0]:
            best = (float(r.fun), r.x.copy(), bool(r.success))
    g, s, ok = best
    nu = np.exp(s)
    psi, f, _ = P.dual_eval_network(net, nu)
    G = nu * (psi + h); G[fixed] = 0.0
    pg = np.where(s <= lo + 1e-12, np.minimum(G, 0.0), np.where(s >= hi - 1e-12, np.maximum(G, 0.0), G))
    r_ = psi + h
    scale = max(1.0, abs(g))
    viol = float(np.where(ct == GE, np.maximum(-r_, 0.0), np.where(ct == EQ, np.abs(r_), 0.0)).max())
    den = max(float(np.abs(psi).max()), float(np.abs(h).max()), 1e-300)
    return dict(value=g, nu=nu, psi=psi, gap=abs(float((nu - c) @ r_)) / scale, infeas=viol / den, pg=float(np.abs(pg).max()) / scale,
                evals=evals[0], lbfgsb_success=ok)
