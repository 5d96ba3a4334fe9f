"""TEST INFRASTRUCTURE -- a THIRD, independent derivation of the shipped instances' optima, in 50-digit arithmetic.

The golden fixtures of round 1 held two derivations per instance (SciPy SLSQP on the primal model; the survey's
dual decomposition + L-BFGS-B), both good to ~1e-6 in the per-pool tenders only.  This module polishes the optimum
to ~1e-30 so that tests can pin tenders at 1e-9: it solves the KKT system of the reference's program
(/root/reference/arbitrage.py:51-78: stationarity of every pool at prices nu, balance of every token that is not
at its price bound, the kink equation of a partially filled constant-sum pool) by Newton's method in mpmath, with

  * every geo-mean pool -- two-asset ones included -- through the n-asset KKT form (x_k = clip(R_k, mu gamma w_k /
    p_k, mu w_k / p_k), sum_k w_k log x_k = sum_k w_k log R_k), NOT the two-asset closed forms the
    C / NumPy / HIP code uses;
  * the active set (which tokens sit on their price bound, which constant-sum pools sit on a kink and in which
    direction) read off an approximate solution, then VERIFIED on the polished point (complementary slackness,
    fills inside (0,1), dual feasibility) -- a wrong guess fails loudly.

PARITY UNPINNED by the reference all the same: none of this is cvxpy output (cvxpy is not installable here).
Only `oracle/make_golden.py` and tests use this file.
"""
import mpmath as mp

mp.mp.dps = 50


def _geomean_pool(R, w, g, p):
    """y = Lambda - Delta of one weighted geo-mean pool at local prices p (all mpf), via the piecewise-linear
    residual F(t) = sum_k w_k f(t - a_k),  a_k = log(R_k p_k / w_k),  f(u) = u (u < 0) | 0 | u + lg (u > -lg)."""
    K = len(R)
    lg = mp.log(g)
    a = [mp.log(R[k] * p[k] / w[k]) for k in range(K)]

    def F(t):
        s = mp.mpf(0)
        for k in range(K):
            u = t - a[k]
            s += w[k] * (u if u < 0 else (u + lg if u > -lg else 0))
        return s
    bps = sorted(set(a + [x - lg for x in a]))
    vals = [F(t) for t in bps]
    t = None
    for i in range(len(bps)):
        if vals[i] == 0:
            t = bps[i]; break
        if i and vals[i - 1] < 0 < vals[i]:
            t = bps[i - 1] - vals[i - 1] * (bps[i] - bps[i - 1]) / (vals[i] - vals[i - 1]); break
    if t is None:
        raise ArithmeticError("geo-mean pool: no root bracket")
    mu = mp.exp(t)
    y = []
    for k in range(K):
        hi = mu * w[k] / p[k]; lo = g * hi
        if R[k] > hi:
            y.append(R[k] - hi)                 # withdrawn
        elif R[k] < lo:
            y.append((R[k] - lo) / g)           # deposited (negative)
        else:
            y.append(mp.mpf(0))
    return y


def _fill_vector(n, l, R, g, sgn):
    d = [mp.mpf(0)] * n
    a, b = l
    if sgn > 0:
        d[a] = -R[1] / g; d[b] = R[1]
    else:
        d[b] = -R[0] / g; d[a] = R[0]
    return d


def polish(inst, nu_approx, tied, tol=mp.mpf(10) ** -40):
    """inst: oracle.instances.normalise()d instance.  nu_approx: prices within ~1e-5 of the optimum.
    tied: {pool index: sgn} of the constant-sum pools sitting on a kink (+1: tender token a, drain b).
    Returns dict(value, nu, psi, y (per pool, pool-local order), theta {pool: fill})."""
    n = inst["n_tokens"]
    L = [list(map(int, l)) for l in inst["local_indices"]]
    R = [[mp.mpf(float(x)) for x in r] for r in inst["reserves"]]
    G = [mp.mpf(float(g)) for g in inst["fees"]]                # (the doubles the scripts compute with: NumPy float64, arbitrage.py:14-28)
    W = []
    for w in inst["weights"]:
        ws = [mp.mpf(float(x)) for x in w]
        tot = sum(ws)
        W.append([x / tot for x in ws])
    kinds = inst["kinds"]
    c = [mp.mpf(float(x)) for x in inst["c"]]
    h = [mp.mpf(float(x)) for x in inst["h"]]
    ctype = [int(x) for x in inst["ctype"]]
    nu0 = [mp.mpf(float(x)) for x in nu_approx]

    # active set from the approximate point
    fixed = [ctype[j] == 2 or (ctype[j] == 0 and c[j] > 0 and nu0[j] <= c[j] * (1 + mp.mpf(10) ** -7)) for j in range(n)]
    tk = sorted(tied)

    def pools_psi(nu):
        psi = [mp.mpf(0)] * n
        ys = []
        for i, l in enumerate(L):
            p = [nu[j] for j in l]
            if kinds[i] == "geomean":
                y = _geomean_pool(R[i], W[i], G[i], p)
            elif i in tied:
                y = [mp.mpf(0), mp.mpf(0)]                      # its fill is an unknown of the system
            else:                                               # constant sum off its kinks: bang-bang
                a, b = p
                if G[i] * b > a:
                    y = [-R[i][1] / G[i], R[i][1]]
                elif G[i] * a > b:
                    y = [R[i][0], -R[i][0] / G[i]]
                else:
                    y = [mp.mpf(0), mp.mpf(0)]
            ys.append(y)
            for k, j in enumerate(l):
                psi[j] += y[k]
        return psi, ys

    free = [j for j in range(n) if not fixed[j]]

    def unpack(z):
        nu = [c[j] if fixed[j] else None for j in range(n)]
        for q, j in enumerate(free):
            nu[j] = mp.exp(z[q])
        th = {i: z[len(free) + q] for q, i in enumerate(tk)}
        return nu, th

    def residual(z):
        nu, th = unpack(z)
        psi, _ = pools_psi(nu)
        r = [psi[j] + h[j] for j in range(n)]
        for i in tk:
            d = _fill_vector(n, L[i], R[i], G[i], tied[i])
            for j in range(n):
                r[j] += th[i] * d[j]
        out = [nu[j] * r[j] for j in free]
        for i in tk:
            a, b = L[i]
            out.append(mp.log(nu[a]) - mp.log(nu[b]) - tied[i] * mp.log(G[i]))
        return out

    z = [mp.log(nu0[j]) for j in free] + [mp.mpf("0.5")] * len(tk)
    m = len(z)
    # tokens no trading pool touches have a flat residual (their price is not unique: two-asset.py's token 1 at some
    # sweep points): they are frozen at the approximate price, which leaves every tender unchanged
    live = list(range(m))
    for it in range(60):
        r = residual(z)
        J = mp.zeros(m, m)
        hstep = mp.mpf(10) ** -25
        for q in range(m):
            zz = list(z); zz[q] += hstep
            rq = residual(zz)
            for p_ in range(m):
                J[p_, q] = (rq[p_] - r[p_]) / hstep
        keep = [q for q in live if any(abs(J[p_, q]) > mp.mpf(10) ** -20 for p_ in range(m))]
        rows = [p_ for p_ in range(m) if any(abs(J[p_, q]) > mp.mpf(10) ** -20 for q in keep)]
        if len(rows) != len(keep):
            raise ArithmeticError(f"KKT system is not square after dropping flat directions: {len(rows)} x {len(keep)}")
        if not keep:
            break
        Js = mp.matrix(len(rows), len(keep))
        for a_, p_ in enumerate(rows):
            for b_, q in enumerate(keep):
                Js[a_, b_] = J[p_, q]
        dz = mp.lu_solve(Js, mp.matrix([-r[p_] for p_ in rows]))
        step = max(abs(x) for x in dz)
        for b_, q in enumerate(keep):
            z[q] += dz[b_]
        if step < tol:
            break
    else:
        raise ArithmeticError("KKT Newton did not converge")
    r = residual(z)
    nu, th = unpack(z)
    psi, ys = pools_psi(nu)
    for i in tk:
        d = _fill_vector(n, L[i], R[i], G[i], tied[i])
        a, b = L[i]
        ys[i] = [th[i] * d[a], th[i] * d[b]]
        for j in range(n):
            psi[j] += th[i] * d[j]
    # ---- verify the guessed active set on the polished point ------------------------------------------------
    eps = mp.mpf(10) ** -25
    for i in tk:
        if not (0 < th[i] < 1):
            raise ArithmeticError(f"pool {i}: fill {th[i]} outside (0,1): wrong active set")
    for j in range(n):
        rj = psi[j] + h[j]
        if ctype[j] == 1 and abs(rj) > eps:
            raise ArithmeticError(f"token {j}: equality residual {rj}")
        if ctype[j] == 0:
            if rj < -eps:
                raise ArithmeticError(f"token {j}: psi + h = {rj} < 0")
            if nu[j] < c[j] - eps:
                raise ArithmeticError(f"token {j}: price below its bound")
            if abs((nu[j] - c[j]) * rj) > eps:
                raise ArithmeticError(f"token {j}: complementary slackness violated ({(nu[j] - c[j]) * rj})")
    value = sum(c[j] * psi[j] for j in range(n))
    dual = sum((nu[j] - c[j]) * h[j] for j in range(n)) + sum(nu[j] * y for i, l in enumerate(L) for j, y in zip(l, ys[i]))
    if abs(dual - value) > mp.mpf(10) ** -20 * max(1, abs(value)):
        raise ArithmeticError(f"duality gap {dual - value}")
    f = lambda x: float(x)
    return dict(value=f(value), value_str=mp.nstr(value, 25), nu=[f(x) for x in nu], psi=[f(x) for x in psi],
                y=[[f(x) for x in y] for y in ys], theta={int(i): f(th[i]) for i in tk},
                residual=f(max([abs(x) for x in r] + [mp.mpf(0)])))
