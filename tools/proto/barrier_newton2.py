"""Prototype 2 (NumPy, CPU): barrier-smoothed dual Newton, general utility (GE/EQ/FREE), 2-asset pools of
kinds cp2 / w2 / curve2 / sum2, safeguarded-Newton per-branch solves.  Design study only."""
import sys, time, argparse
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np
import scipy.linalg as sla
from cfmm import synthetic
import cfmm

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=0.02); ap.add_argument("--config", default="C5")
ap.add_argument("--sigma", type=float, default=0.2); ap.add_argument("--util", default="liq")
ap.add_argument("--seed", type=int, default=0); ap.add_argument("--sum2", type=int, default=0)
ap.add_argument("-q", action="store_true")
a_ = ap.parse_args()
net = synthetic.config(a_.config, scale=a_.scale, seed=a_.seed)
n = net["n_tokens"]
rng = np.random.default_rng(1 + a_.seed)
if a_.util == "liq":
    h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
    tgt = int(rng.integers(0, n)); h[tgt] = 0
    u = cfmm.Liquidate(h, tgt)
elif a_.util == "swap":
    h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
    tgt = int(rng.integers(0, n)); h[tgt] = 0
    u = cfmm.Swap(h, tgt)
else:
    u = cfmm.Arbitrage(net["c"])
c, h, ctype = np.asarray(u.c, float), np.asarray(u.h, float), np.asarray(u.ctype)

KCP, KW, KCV, KSUM = 0, 1, 2, 3
cols = dict(Ri=[], Ro=[], fee=[], par=[], par2=[], ti=[], to=[], kind=[])
def add(kind, Ra, Rb, fee, ia, ib, pa=None, pb=None):
    m = len(Ra); z = np.zeros(m)
    cols["Ri"] += [Ra, Rb]; cols["Ro"] += [Rb, Ra]; cols["fee"] += [fee, fee]
    cols["par"] += [z if pa is None else pa, z if pb is None else pb]
    cols["ti"] += [ia, ib]; cols["to"] += [ib, ia]; cols["kind"] += [np.full(2 * m, kind)]
if "cp2" in net: b = net["cp2"]; add(KCP, b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"])
if "w2" in net: b = net["w2"]; add(KW, b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"], b["wa"] / (1 - b["wa"]), (1 - b["wa"]) / b["wa"])
if "curve2" in net: b = net["curve2"]; add(KCV, b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"], b["alpha"], b["alpha"])
if a_.sum2:
    m = a_.sum2; ia = rng.integers(0, n, m); ib = (ia // 4) * 4 + (ia % 4 + rng.integers(1, 4, m)) % 4 if a_.config == "C5" else (ia + rng.integers(1, n, m)) % n
    L = np.exp(rng.normal(np.log(1e3), 1.5, m)); pi = net["prices"]
    # constant-sum pools quote a fixed 1:1 rate: only sensible between equal-priced tokens -> rescale units
    add(KSUM, L / pi[ia], L / pi[ia], np.full(m, 0.999), ia.astype(np.int32), ib.astype(np.int32))
Ri, Ro, fee, par, ti, to, kind = (np.concatenate(cols[k]) for k in ("Ri", "Ro", "fee", "par", "ti", "to", "kind"))
NB = len(Ri)
kp = Ri * Ro
Kc = Ri + Ro - np.where(kind == KCV, par / kp, 0.0)
cap = np.where(kind == KSUM, Ro / fee, np.inf)

def lam(D):
    """L(D), L'(D), L''(D) of each branch's forward exchange function."""
    x = Ri + fee * D
    with np.errstate(all='ignore'):
        # constant product
        Lp = Ro - kp / x; L1p = fee * kp / (x * x); L2p = -2 * fee * fee * kp / x ** 3
        # weighted: L = Ro (1 - (Ri/x)^r), r = w_in / w_out
        r = par; q = (Ri / x) ** r
        Lw = Ro * (1 - q); L1w = fee * Ro * r * q / x; L2w = -fee * fee * Ro * r * (r + 1) * q / (x * x)
        # curve
        al = par; b = Kc - x
        Y = 0.5 * (b + np.sqrt(b * b + 4 * al / x))
        fx = 1 + al / (x * x * Y); fy = 1 + al / (x * Y * Y); Y1 = -fx / fy
        fxx = -2 * al / (x ** 3 * Y); fxy = -al / (x * x * Y * Y); fyy = -2 * al / (x * Y ** 3)
        Y2 = -(fxx + 2 * fxy * Y1 + fyy * Y1 * Y1) / fy
        Lc = Ro - Y; L1c = -fee * Y1; L2c = -fee * fee * Y2
    L = np.select([kind == KCP, kind == KW, kind == KCV], [Lp, Lw, Lc], fee * D)
    L1 = np.select([kind == KCP, kind == KW, kind == KCV], [L1p, L1w, L1c], fee)
    L2 = np.select([kind == KCP, kind == KW, kind == KCV], [L2p, L2w, L2c], 0.0)
    return L, L1, L2

L0, L10, L20 = lam(np.zeros(NB))
inner_iters = []; hist = []
def branch(nu, mu):
    ni = nu[ti]; no = nu[to]
    issum = kind == KSUM
    def F(D):
        L, L1, L2 = lam(D)
        f = no * L1 - ni; fd = no * L2
        if mu > 0:
            f = f + mu / D; fd = fd - mu / (D * D)
            f = np.where(issum, f - mu / (cap - D), f); fd = np.where(issum, fd - mu / (cap - D) ** 2, fd)
        return f, fd, L, L1
    if mu == 0:
        # exact: bisection on the interior root (reference semantics), gate by the band
        F0 = no * L10 - ni
        lo = np.full(NB, -60.0) + np.log(Ri); hi = np.full(NB, 40.0) + np.log(Ri)
        for _ in range(80):
            mid = 0.5 * (lo + hi); f, _, _, _ = F(np.exp(mid)); pos = f > 0
            lo = np.where(pos, mid, lo); hi = np.where(pos, hi, mid)
        D = np.where(F0 > 0, np.exp(0.5 * (lo + hi)), 0.0)
        D = np.where(issum, np.where(F0 > 0, cap, 0.0), D)
        L, L1, L2 = lam(D)
        kappa = np.where((D > 0) & ~issum, -1.0 / (no * L2 - 1e-300), 0.0)
        return D, np.where(D > 0, L, 0.0), kappa, L1
    # smoothed: start from the root of the local quadratic model, safeguarded Newton in D
    F0 = no * L10 - ni; c2 = np.maximum(-no * L20, 1e-300)
    rt = np.sqrt(F0 * F0 + 4 * c2 * mu)
    D = np.where(issum, 0.0, np.where(F0 > 0, (F0 + rt) / (2 * c2), 2 * mu / (rt - F0)))
    # constant sum: closed form root of  -s D^2 + (s cap - 2 mu) D + mu cap = 0
    sgap = fee * no - ni
    with np.errstate(all='ignore'):
        bq = sgap * cap - 2 * mu
        Dsum = np.where(np.abs(sgap) * cap > 1e-9 * mu, (bq + np.sqrt(bq * bq + 4 * sgap * mu * cap) * 1.0) / (2 * sgap), cap / 2)
        # stable form for sgap<0: D = 2 mu cap / (-bq + sqrt(...))
        Dsum2 = 2 * mu * cap / (-bq + np.sqrt(bq * bq + 4 * sgap * mu * cap))
        Dsum = np.where(bq > 0, Dsum, Dsum2)
    D = np.where(issum, Dsum, D)
    # start: exact interior root where it is closed form and positive
    with np.errstate(all='ignore'):
        xe_cp = np.sqrt(fee * no * kp / ni)
        r = par
        xe_w = Ri * (fee * no * Ro * r / (ni * Ri)) ** (1.0 / (r + 1.0))
        De = np.select([kind == KCP, kind == KW], [(xe_cp - Ri) / fee, (xe_w - Ri) / fee], 0.0)
    D = np.where(~issum & (De > 0), De, np.where(issum, D, 0.0))
    lo = np.zeros(NB); hi = np.where(issum, cap, np.inf)
    done = issum.copy(); its = 0; nits = np.zeros(NB); dprev = np.full(NB, np.inf)
    for it in range(100):
        L, L1, L2 = lam(D)
        A = no * L1 - ni; A1 = np.minimum(no * L2, -1e-300)
        with np.errstate(all='ignore'):
            f = A + np.where(D > 0, mu / D, np.inf)
        lo = np.where(~done & (f > 0), D, lo); hi = np.where(~done & (f <= 0), D, hi)
        bq = A - A1 * D
        rt = np.sqrt(bq * bq - 4 * A1 * mu)
        with np.errstate(all='ignore'):
            Dn = np.where(bq > 0, (bq + rt) / (-2 * A1), 2 * mu / (rt - bq))
        convc = np.abs(Dn - D) <= 1e-13 * np.maximum(Dn, D)
        ok = convc | ((Dn > lo) & (Dn < hi) & (np.abs(Dn - D) < 0.5 * dprev))
        Dn = np.where(ok, Dn, np.where(np.isfinite(hi), 0.5 * (lo + hi), 2 * D + 1e-300))
        step = np.abs(Dn - D)
        conv = step <= 1e-13 * np.maximum(Dn, D)
        dprev = np.where(done, dprev, step)
        nits = np.where(done, nits, it + 1)
        D = np.where(done, D, Dn)
        done = done | conv; its += 1
        if done.all(): break
    inner_iters.append(its); hist.append(np.bincount(nits.astype(int), minlength=101))
    f, fd, L, L1 = F(D)
    kappa = -1.0 / fd
    return D, L, kappa, L1

ge = ctype == 0; eq = ctype == 1; fr = ctype == 2
def evaluate(s, mu, hess=False):
    nu = np.exp(s)
    D, L, kappa, L1 = branch(nu, mu)
    psi = np.bincount(to, L, n) - np.bincount(ti, D, n)
    trade = (nu[to] * L - nu[ti] * D).sum()
    g = (nu - c) @ h + trade
    G = nu * (psi + h)
    Hd = np.maximum(G, 0)
    if mu > 0:
        g += mu * (np.log(D).sum() + np.log((cap - D)[kind == KSUM]).sum())
        slack = nu[ge] - c[ge]
        g -= mu * np.log(slack).sum()
        G[ge] -= mu * nu[ge] / slack
        Hd[ge] = np.maximum(G[ge], 0) + mu * nu[ge] * c[ge] / slack ** 2    # d/ds of -mu nu/(nu-c) = mu nu c/(nu-c)^2
    out = dict(g=g, G=G, psi=psi, nu=nu, trade=trade)
    if hess:
        wi = nu[ti]; wo = -L1 * nu[to]
        H = np.zeros((n, n))
        np.add.at(H, (ti, ti), kappa * wi * wi); np.add.at(H, (to, to), kappa * wo * wo)
        np.add.at(H, (ti, to), kappa * wi * wo); np.add.at(H, (to, ti), kappa * wi * wo)
        H[np.diag_indices(n)] += Hd
        out["H"] = H
    return out

def certs(s, e):
    ex = evaluate(s, 0.0)
    nu = ex["nu"]; dual = ex["g"]
    r = e["psi"] + h
    viol = np.where(eq, np.abs(r), np.where(ge, np.maximum(-r, 0), 0)).max(); sc = max(np.abs(e["psi"]).max(), np.abs(h).max(), 1e-300)
    sub = ex["trade"] - e["trade"]
    cs = (nu - c) @ r
    return (sub + cs) / max(1, abs(dual)), viol / sc, dual, c @ e["psi"], sub, cs

nu0 = cfmm.start_prices(net, u)
s = np.log(nu0)
s[fr] = np.log(c[fr])
s[ge] = np.maximum(s[ge], np.log(np.maximum(c[ge], 1e-300)) + 1e-3)
free = ~fr
nf = int(free.sum())
ex = evaluate(s, 0.0)
NBAR = NB + int(ge.sum())
mu = 0.1 * abs(ex["g"]) / NBAR
if not a_.q: print("branches", NB, "g0", ex["g"], "mu0", mu)
t0 = time.time(); nev = 0; endgame = False
for it in range(300):
    e = evaluate(s, mu, True); nev += 1
    G = e["G"]; H = e["H"]
    gap, inf, dual, primal, sub, cs = certs(s, e)
    Hr = H[np.ix_(free, free)]
    reg = 0.0
    while True:
        try:
            cf = sla.cho_factor(Hr + reg * np.eye(nf)); break
        except Exception:
            reg = max(1e-12 * np.trace(Hr) / nf, reg * 100)
    d = np.zeros(n); d[free] = -sla.cho_solve(cf, G[free])
    dec = -(G @ d)
    if not a_.q:
        print("%3d ev %3d mu %.2e g_mu %.10g dual %.10g primal %.10g gap %.2e (sub %.1e cs %.1e) infeas %.2e dec %.2e |d| %.2e reg %.0e in %d"
          % (it, nev, mu, e["g"], dual, primal, gap, sub, cs, inf, dec, np.abs(d).max(), reg, inner_iters[-2] if len(inner_iters) > 1 else 0))
    if abs(gap) <= 1e-6 and inf <= 1e-6:
        break
    # fraction to the boundary for the GE barrier (s > log c)
    t = min(1.0, 2.0 / max(np.abs(d).max(), 1e-300))
    room = s - np.log(np.maximum(c, 1e-300)); m_ = ge & (d < 0) & (c > 0)
    if m_.any(): t = min(t, 0.9 * np.min(room[m_] / -d[m_]))
    t_first = t
    for ls in range(40):
        e2 = evaluate(s + t * d, mu); nev += 1
        if e2["g"] <= e["g"] - 1e-4 * t * dec or (dec < 1e-13 * abs(e["g"])): break
        t *= 0.5
    s = s + t * d
    if abs(gap) <= 1e-6:
        pass                                    # gap is there: finish centering at this mu
    elif dec < 10 * mu * NBAR and (t == t_first or dec < 1e-3 * mu * NBAR):
        mu *= a_.sigma
print("inner hist", np.sum(hist, axis=0)); print("RESULT %s scale %g util %s seed %d sum2 %d: newton %d evals %d gap %.1e infeas %.1e dual %.8g time %.1f inner max %d"
      % (a_.config, a_.scale, a_.util, a_.seed, a_.sum2, it, nev, gap, inf, dual, time.time() - t0, max(inner_iters)))
