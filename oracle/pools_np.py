"""TEST INFRASTRUCTURE -- CPU oracle, part 1: per-pool arbitrage subproblems (NumPy, fp64).

PARITY UNPINNED: the reference (/root/reference, four cvxpy scripts) holds no expected
outputs and cvxpy is not installable here; this file is pinned instead against
(i) the SciPy primal NLP of oracle/primal_scipy.py, which states the reference model
verbatim, and (ii) the survey-derived known answers in tests/golden/.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may use oracle/.

For given local prices p > 0 every pool solves, independently,

    arb_i(p) = max  p'(L - D)
               s.t. phi_i(R + gamma*D - L) >= phi_i(R),  D, L >= 0

which is the per-pool piece of the reference's model: variables
/root/reference/arbitrage.py:51-52, post-trade reserves :60, trading-function constraints
:63-74 (weighted geo-mean :65, Uniswap v2 :68-70, constant sum :73-74).  Every function
returns y = L - D (pool-local order; negative = tendered, positive = received) and the
value arb = p'y.  By the envelope theorem y is the gradient of arb_i wrt p.
"""
import numpy as np


# --------------------------------------------------------------------------------------
# weighted geometric mean, 2 assets (constant product is wa == wb)      arbitrage.py:65,68
# --------------------------------------------------------------------------------------
def arb_geomean2(Ra, Rb, gamma, wa, wb, pa, pb):
    """Vectorised over pools.  Direction a->b (tender a, receive b) is active iff
    gamma*wa*pb*Rb > wb*pa*Ra; b->a iff gamma*wb*pa*Ra > wa*pb*Rb; never both (gamma<=1).
    With eta = w_in/w_out the new reserve of the tendered token is
        x = (gamma * eta * (p_out/p_in) * R_out * R_in**eta) ** (1/(eta+1))
    """
    Ra, Rb, gamma, wa, wb, pa, pb = map(np.asarray, (Ra, Rb, gamma, wa, wb, pa, pb))
    ya = np.zeros(np.broadcast(Ra, pa).shape)
    yb = np.zeros_like(ya)

    def one_dir(Rin, Rout, win, wout, pin, pout):
        eta = win / wout
        act = gamma * win * pout * Rout > wout * pin * Rin
        with np.errstate(all="ignore"):
            x = np.exp((np.log(gamma * eta * (pout / pin) * Rout) + eta * np.log(Rin)) / (eta + 1.0))
            x = np.where(eta == 1.0, np.sqrt(gamma * (pout / pin) * Rin * Rout), x)
            d_in = (x - Rin) / gamma
            l_out = np.where(eta == 1.0, Rout - Rin * Rout / x, Rout * (1.0 - (Rin / x) ** eta))
        return act, np.where(act, -d_in, 0.0), np.where(act, l_out, 0.0)

    act_ab, ya_ab, yb_ab = one_dir(Ra, Rb, wa, wb, pa, pb)
    act_ba, yb_ba, ya_ba = one_dir(Rb, Ra, wb, wa, pb, pa)
    ya = ya_ab + ya_ba
    yb = yb_ab + yb_ba
    return ya, yb, pa * ya + pb * yb


# --------------------------------------------------------------------------------------
# weighted geometric mean, n assets                                  arbitrage.py:65
# --------------------------------------------------------------------------------------
def arb_geomean_n(R, w, gamma, p):
    """One pool, n_i assets, w normalised.  KKT with multiplier mu:
        x_k(mu) = clip(R_k, mu*gamma*w_k/p_k, mu*w_k/p_k)
    and sum_k w_k log x_k(mu) = sum_k w_k log R_k.  In t = log mu, a_k = log(R_k p_k/w_k),
    lg = log gamma <= 0 the residual is the piecewise-linear non-decreasing function
        F(t) = sum_k w_k f(t - a_k),  f(u) = u (u<0) | 0 (0<=u<=-lg) | u+lg (u>-lg)
    solved exactly by scanning its 2 n_i breakpoints.
    """
    R = np.asarray(R, float); w = np.asarray(w, float); p = np.asarray(p, float)
    lg = np.log(gamma)
    a = np.log(R * p / w)

    def F(t):
        u = t - a
        return float(np.sum(w * np.where(u < 0, u, np.where(u > -lg, u + lg, 0.0))))

    bps = np.sort(np.concatenate([a, a - lg]))
    Fv = np.array([F(t) for t in bps])
    # no-trade plateau: some breakpoint interval where F == 0 with everything in the dead zone
    lo_i = np.where(Fv <= 0)[0].max()         # F(bps[0]) <= 0 always
    hi_i = np.where(Fv >= 0)[0].min()         # F(bps[-1]) >= 0 always
    if Fv[lo_i] == 0.0 or Fv[hi_i] == 0.0 or hi_i <= lo_i:
        t = bps[lo_i] if Fv[lo_i] == 0.0 else bps[hi_i]
    else:
        t0, t1, f0, f1 = bps[lo_i], bps[hi_i], Fv[lo_i], Fv[hi_i]
        t = t0 - f0 * (t1 - t0) / (f1 - f0)
    mu = np.exp(t)
    x = np.clip(R, mu * gamma * w / p, mu * w / p)
    y = np.where(x < R, R - x, (R - x) / gamma)
    return y, float(p @ y)


def arb_geomean_n_vec(R, w, gamma, p):
    """arb_geomean_n for a whole bucket at once: R, w, p are [k][m] (slot-major, as cfmm.pack lays a size class out), gamma [m].
    The same exact piecewise-linear root, its 2k breakpoints per pool evaluated as one [2k][k][m] array expression -- the
    "vectorised NumPy, one thread" CPU baseline of BASELINE.md section 4 (bench.py: cpu_baseline.numpy_one_thread); pinned
    against the per-pool form in tests/test_oracle.py.  Returns y [k][m] and the pools' values [m]."""
    R = np.asarray(R, float); w = np.asarray(w, float); p = np.asarray(p, float); gamma = np.asarray(gamma, float)
    lg = np.log(gamma)[None, :]
    a = np.log(R * p / w)
    bps = np.sort(np.concatenate([a, a - lg], axis=0), axis=0)                      # [2k][m]
    u = bps[:, None, :] - a[None, :, :]                                              # [2k][k][m]
    Fv = np.sum(w[None] * np.where(u < 0, u, np.where(u > -lg[None], u + lg[None], 0.0)), axis=1)       # [2k][m]
    q = np.arange(bps.shape[0])[:, None]
    lo_i = np.max(np.where(Fv <= 0, q, 0), axis=0)
    hi_i = np.min(np.where(Fv >= 0, q, bps.shape[0] - 1), axis=0)
    cols = np.arange(bps.shape[1])
    t0, t1, f0, f1 = bps[lo_i, cols], bps[hi_i, cols], Fv[lo_i, cols], Fv[hi_i, cols]
    flat = (f0 == 0.0) | (f1 == 0.0) | (hi_i <= lo_i)
    with np.errstate(all="ignore"):
        t = np.where(flat, np.where(f0 == 0.0, t0, t1), t0 - f0 * (t1 - t0) / np.where(f1 == f0, 1.0, f1 - f0))
    mu = np.exp(t)[None, :]
    x = np.clip(R, mu * gamma[None, :] * w / p, mu * w / p)
    y = np.where(x < R, R - x, (R - x) / gamma[None, :])
    return y, np.sum(p * y, axis=0)


def dual_eval_network(net, nu):
    """psi(nu) and sum_i arb_i(nu) of a whole synthetic network of the reference's pool kinds (cfmm/synthetic.py's SoA buckets: cp2, w2,
    gn) as vectorised NumPy -- the evaluation bench.py times as its one-thread baseline; pinned against the C twin in tests/test_oracle.py"""
    if set(net) & {"curve2", "pow2", "gk", "sum2"}:
        raise ValueError("dual_eval_network: constant-product / weighted / n-asset geometric-mean buckets only")
    n = net["n_tokens"]
    psi = np.zeros(n); arb = 0.0; m = 0
    for key in ("cp2", "w2"):
        if key not in net:
            continue
        b = net[key]
        pa, pb = nu[b["ia"]], nu[b["ib"]]
        wa = b["wa"] if key == "w2" else 0.5
        ya, yb, v = arb_geomean2(b["Ra"], b["Rb"], b["fee"], wa, 1.0 - wa, pa, pb)
        psi += np.bincount(b["ia"], weights=ya, minlength=n) + np.bincount(b["ib"], weights=yb, minlength=n)
        arb += float(np.sum(v)); m += len(b["Ra"])
    for k, b in net.get("gn", {}).items():
        y, v = arb_geomean_n_vec(b["R"], b["w"], b["fee"], nu[b["idx"]])
        psi += np.bincount(b["idx"].ravel(), weights=y.ravel(), minlength=n)
        arb += float(np.sum(v)); m += b["R"].shape[1]
    return psi, arb, m


# --------------------------------------------------------------------------------------
# constant sum                                                       arbitrage.py:73-74
# --------------------------------------------------------------------------------------
def arb_sum(R, gamma, p):
    """LP: tender the cheapest token a = argmin p; withdraw every b with gamma*p_b > p_a
    completely.  Piecewise linear in p: on the kink gamma*p_b == p_a any fill fraction of
    that leg is optimal (handled by the caller's primal recovery)."""
    R = np.asarray(R, float); p = np.asarray(p, float)
    a = int(np.argmin(p))
    y = np.zeros_like(R)
    take = gamma * p > p[a]
    take[a] = False
    y[take] = R[take]
    y[a] = -R[take].sum() / gamma
    return y, float(p @ y)


# --------------------------------------------------------------------------------------
# Curve-style 2-asset pool: phi(x, y) = x + y - alpha/(x*y)            (not in reference)
# --------------------------------------------------------------------------------------
def curve_alpha_from_A(Ra, Rb, A):
    """alpha such that phi_alpha(x) >= phi_alpha(R) is the on-chain 2-coin StableSwap
    invariant D(x) >= D(R):  4A(x+y) + D = 4AD + D^3/(4xy)  <=>  x + y - (D^3/16A)/(xy) = D(1-1/4A)."""
    S = Ra + Rb
    D = S
    for _ in range(64):
        f = 4 * A * S + D - 4 * A * D - D ** 3 / (4 * Ra * Rb)
        df = 1 - 4 * A - 3 * D ** 2 / (4 * Ra * Rb)
        Dn = D - f / df
        if abs(Dn - D) <= 1e-15 * D:
            D = Dn
            break
        D = Dn
    return D ** 3 / (16 * A)


def _curve_y(x, C, alpha):
    # positive root of  x y^2 + (x^2 - C x) y - alpha = 0
    b = C - x
    return 0.5 * (b + np.sqrt(b * b + 4.0 * alpha / x))


def arb_curve2(Ra, Rb, gamma, alpha, pa, pb, iters=60):
    """One pool.  Marginal price of a in units of b at reserves (x,y):
        m(x,y) = phi_x/phi_y = (1 + alpha/(x^2 y)) / (1 + alpha/(x y^2)).
    Tender a iff m(R) > pa/(gamma pb); then solve m(x, y(x)) = pa/(gamma pb) for x > Ra
    (bisection here: the oracle favours certainty over speed)."""
    C = Ra + Rb - alpha / (Ra * Rb)

    def solve(Rin, Rout, pin, pout):
        rho = pin / (gamma * pout)
        m0 = (1 + alpha / (Rin * Rin * Rout)) / (1 + alpha / (Rin * Rout * Rout))
        if not m0 > rho:
            return None

        def h(x):
            y = _curve_y(x, C, alpha)
            return (1 + alpha / (x * x * y)) / (1 + alpha / (x * y * y)) - rho
        lo, hi = Rin, Rin * 2.0
        while h(hi) > 0:
            hi *= 2.0
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            if h(mid) > 0:
                lo = mid
            else:
                hi = mid
            if hi - lo <= 1e-16 * hi:
                break
        x = 0.5 * (lo + hi)
        return -(x - Rin) / gamma, Rout - _curve_y(x, C, alpha)

    s = solve(Ra, Rb, pa, pb)
    if s is not None:
        ya, yb = s
    else:
        s = solve(Rb, Ra, pb, pa)
        if s is not None:
            yb, ya = s
        else:
            ya = yb = 0.0
    return np.array([ya, yb]), float(pa * ya + pb * yb)


def arb_power2(Ra, Rb, gamma, t, pa, pb):
    """Power-sum pool  phi(x, y) = x^(1-t) + y^(1-t),  0 < t < 1  (YieldSpace's curve; not in the reference -- the first
    tenant of the library's GENERIC two-asset bucket, include/cfmm.h CFMM_POOL_POW2).  CLOSED FORM, which the library's
    exact path deliberately does not use (it finds the same root by a safeguarded Newton search on the table entry's L'):
    marginal price of a in units of b at reserves (x, y):  m = phi_x / phi_y = (y / x)^t.  Tender a iff
    gamma pb m(R) > pa; at the optimum (y / x)^t = rho := pa / (gamma pb), on the level set x^q + y^q = K, q = 1 - t:
        x = (K / (1 + rho^(q/t)))^(1/q),   y = x rho^(1/t).
    Vectorised over pools."""
    Ra, Rb, gamma, t, pa, pb = (np.asarray(v, float) for v in (Ra, Rb, gamma, t, pa, pb))
    q = 1.0 - t

    def one_dir(Rin, Rout, pin, pout):
        # (in logarithms of x / R_in and y / R_out: R - x formed directly loses the trade of a pool whose reserves differ by
        #  ten orders of magnitude -- and its profit, a difference of the two legs' values -- to cancellation)
        trade = gamma * pout * (Rout / Rin) ** t > pin
        lrho = np.log(pin / (gamma * pout))
        a, bb = (Rout / Rin) ** q, np.exp(lrho * q / t)
        lx = np.log1p((a - bb) / (1.0 + bb)) / q               # log(x / R_in),  x^q = K / (1 + rho^(q/t))  [the price condition]
        # what is received follows from the level set, not from y = x rho^(1/t): out of a reserve 1e10 times the other one
        # log(y / R_out) is ~1e-12, and lx - log(R_out / R_in) + log(rho) / t would keep five digits of it
        z = np.expm1(q * lx) / a                               # (x^q - R_in^q) / R_out^q
        ly = np.log1p(-z) / q
        return trade, np.where(trade, -Rin * np.expm1(lx) / gamma, 0.0), np.where(trade, -Rout * np.expm1(ly), 0.0)
    tab, ya1, yb1 = one_dir(Ra, Rb, pa, pb)
    tba, yb2, ya2 = one_dir(Rb, Ra, pb, pa)
    ya = np.where(tab, ya1, np.where(tba, ya2, 0.0))
    yb = np.where(tab, yb1, np.where(tba, yb2, 0.0))
    return ya, yb, pa * ya + pb * yb


# --------------------------------------------------------------------------------------
# n-asset stableswap  phi(x) = sum x - alpha / prod x                  (not in reference; the K-asset table's first
# smooth tenant: csrc/phik.hpp PhiK<0>; K = 2 is arb_curve2's function)
# --------------------------------------------------------------------------------------
def arb_stable_n(R, alpha, gamma, p, outer=80, inner=64):
    """Vectorised over pools: R, p are [k, m], alpha, gamma [m].  KKT with multiplier mu and coupling s = alpha / prod x:
    phi_j = 1 + s / x_j, so x_j = clip(R_j, s / (p_j / (gamma mu) - 1), s / (p_j / mu - 1)); nested bisection on
    (log mu, log s) -- the restatement of pool_generic_k (same two scalar equations, NumPy arithmetic)."""
    R = np.asarray(R, float); p = np.asarray(p, float)
    k, m = R.shape
    alpha = np.broadcast_to(np.asarray(alpha, float), (m,)); gamma = np.broadcast_to(np.asarray(gamma, float), (m,))
    lR = np.log(R); la = np.log(alpha)
    lsR = la - lR.sum(axis=0); sR = np.exp(lsR)
    lmax = np.log(p.min(axis=0) / gamma)

    def legs(lm, ls):
        with np.errstate(all="ignore"):
            qd = p * np.exp(-lm)[None, :] / gamma[None, :]; qw = p * np.exp(-lm)[None, :]
            gd = np.where(qd > 1.0, np.log(np.maximum(qd - 1.0, 1e-300)), -np.inf)
            gw = np.where(qw > 1.0, np.log(np.maximum(qw - 1.0, 1e-300)), -np.inf)
        return np.clip(lR, ls[None, :] - gd, ls[None, :] - gw), np.isinf(gd).any(axis=0)

    def coupling(lm):
        a = lsR - 90.0; b = lsR + 90.0
        for _ in range(inner):
            ls = 0.5 * (a + b)
            lx, _open = legs(lm, ls)
            hi = ls - (la - lx.sum(axis=0)) > 0.0
            b = np.where(hi, ls, b); a = np.where(hi, a, ls)
        ls = 0.5 * (a + b)
        lx, op = legs(lm, ls)
        return ls, lx, op

    def gap(lm):
        ls, lx, op = coupling(lm)
        dx = np.where(lx == lR, 0.0, np.exp(np.minimum(lx, 700.0)) - R).sum(axis=0)
        return np.where(op, 1.0, dx - (np.exp(ls) - sR)), lx

    lo = lmax - 90.0; hi = lmax.copy()
    for _ in range(outer):
        lm = 0.5 * (lo + hi)
        v, _ = gap(lm)
        neg = v < 0.0
        lo = np.where(neg, lm, lo); hi = np.where(neg, hi, lm)
    _, lx = gap(hi)
    d = np.where(lx == lR, 0.0, R - np.exp(lx))
    y = np.where(d > 0.0, d, d / gamma[None, :])
    return y, (p * y).sum(axis=0)


def arb_pool_primal(R, gamma, p, phi, dphi, x_floor=1e-9):
    """ONE pool's arbitrage subproblem as the reference writes it (arbitrage.py:51-52,60,63-74), by SLSQP: independent of
    every dual-side solver.  phi / dphi: the trading function and its gradient on the new reserves."""
    from scipy.optimize import minimize
    R = np.asarray(R, float); p = np.asarray(p, float)
    k = len(R)
    scale = float(p @ R)

    def newres(z):
        return R + gamma * z[:k] - z[k:]
    cons = [dict(type="ineq", fun=lambda z: (phi(np.maximum(newres(z), x_floor * R)) - phi(R)) / R.sum(),
                 jac=lambda z: np.concatenate([gamma * dphi(np.maximum(newres(z), x_floor * R)), -dphi(np.maximum(newres(z), x_floor * R))]) / R.sum()),
            dict(type="ineq", fun=lambda z: (newres(z) - x_floor * R) / R, jac=lambda z: np.hstack([gamma * np.eye(k), -np.eye(k)]) / R[:, None])]
    best = None
    for z0 in (np.zeros(2 * k), np.concatenate([0.1 * R, 0.1 * R])):
        res = minimize(lambda z: -float(p @ (z[k:] - z[:k])) / scale, z0, jac=lambda z: -np.concatenate([-p, p]) / scale,
                       bounds=[(0, None)] * (2 * k), constraints=cons, method="SLSQP", options=dict(ftol=1e-16, maxiter=500))
        if best is None or res.fun < best.fun:
            best = res
    z = best.x
    # (Delta_j and Lambda_j both positive on one leg is never optimal with gamma < 1, but SLSQP may leave a little of it)
    y = z[k:] - z[:k]
    return y, float(p @ y)
