"""Prototype (NumPy, CPU): barrier-smoothed dual Newton for near-linear 2-asset pools (config 5).
Not product code: a design study for the second-order outer iteration."""
import sys, time
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np
import scipy.linalg as sla
from cfmm import synthetic
import cfmm

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
cfgname = sys.argv[2] if len(sys.argv) > 2 else "C5"
sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
net = synthetic.config(cfgname, scale=scale)
n = net["n_tokens"]
rng = np.random.default_rng(1)
h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
tgt = int(rng.integers(0, n)); h[tgt] = 0
u = cfmm.Liquidate(h, tgt)

# directed branches: (Ri, Ro, fee, alpha or 0, i, o)
Ri = []; Ro = []; fee = []; al = []; ti = []; to = []
for key in ("cp2", "curve2"):
    if key in net:
        b = net[key]; m = len(b["Ra"]); a = b["alpha"] if key == "curve2" else np.zeros(m)
        Ri += [b["Ra"], b["Rb"]]; Ro += [b["Rb"], b["Ra"]]; fee += [b["fee"]] * 2; al += [a] * 2
        ti += [b["ia"], b["ib"]]; to += [b["ib"], b["ia"]]
Ri, Ro, fee, al, ti, to = map(np.concatenate, (Ri, Ro, fee, al, ti, to))
NB = len(Ri)
Kc = Ri + Ro - np.where(al > 0, al / (Ri * Ro), 0.0)       # curve level
kp = Ri * Ro
iscurve = al > 0

def curve(x):
    """Y, p=-Y', Y'' on the pool's level set."""
    with np.errstate(all='ignore'):
        b = Kc - x
        Yc = 0.5 * (b + np.sqrt(b * b + 4 * al / x))
        fx = 1 + al / (x * x * Yc); fy = 1 + al / (x * Yc * Yc)
        Y1 = -fx / fy
        fxx = -2 * al / (x ** 3 * Yc); fxy = -al / (x * x * Yc * Yc); fyy = -2 * al / (x * Yc ** 3)
        Y2c = -(fxx + 2 * fxy * Y1 + fyy * Y1 * Y1) / fy
        Yp = kp / x
    Y = np.where(iscurve, Yc, Yp)
    p = np.where(iscurve, -Y1, kp / (x * x))
    Y2 = np.where(iscurve, Y2c, 2 * kp / x ** 3)
    return Y, p, Y2

def branch(nu, mu):
    """smoothed (mu>0) or exact (mu=0) optimal tender per branch. returns D, L, kappa, gp"""
    ni = nu[ti]; no = nu[to]; a = no * fee
    def F(D):
        _, p, _ = curve(Ri + fee * D)
        return a * p - ni + (mu / D if mu > 0 else 0.0)
    lo = np.full(NB, -80.0) + np.log(Ri); hi = np.full(NB, 40.0) + np.log(Ri)
    for _ in range(70):
        mid = 0.5 * (lo + hi); f = F(np.exp(mid))
        pos = f > 0
        lo = np.where(pos, mid, lo); hi = np.where(pos, hi, mid)
    D = np.exp(0.5 * (lo + hi))
    if mu == 0:
        _, p0, _ = curve(Ri)
        D = np.where(a * p0 - ni > 0, D, 0.0)
    Y, p, Y2 = curve(Ri + fee * D)
    L = Ro - Y
    L = np.where(D > 0, L, 0.0)
    FD = -a * fee * Y2 - (mu / np.maximum(D, 1e-300) ** 2 if mu > 0 else 0.0)
    kappa = np.where(D > 0, -1.0 / FD, 0.0)
    return D, L, kappa, fee * p

def evaluate(s, mu, hess=False):
    nu = np.exp(s)
    D, L, kappa, gp = branch(nu, mu)
    psi = np.bincount(to, L, n) - np.bincount(ti, D, n)
    val = nu[to] * L - nu[ti] * D
    if mu > 0:
        val = val + mu * np.log(D)
    g = nu @ h + val.sum()
    G = nu * (psi + h)
    out = dict(g=g, G=G, psi=psi, nu=nu, tradeval=(nu[to] * L - nu[ti] * D).sum())
    if hess:
        wi = nu[ti]; wo = -gp * nu[to]
        H = np.zeros((n, n))
        np.add.at(H, (ti, ti), kappa * wi * wi); np.add.at(H, (to, to), kappa * wo * wo)
        np.add.at(H, (ti, to), kappa * wi * wo); np.add.at(H, (to, ti), kappa * wi * wo)
        out["H"] = H
    return out

def certs(s, psi_mu, trade_mu):
    ex = evaluate(s, 0.0)
    nu = ex["nu"]
    dual = ex["g"]
    primal = psi_mu[tgt]
    r = psi_mu + h
    viol = np.abs(np.delete(r, tgt)).max(); sc = max(np.abs(psi_mu).max(), np.abs(h).max())
    # gap = pool suboptimality + complementary slackness
    sub = ex["tradeval"] - trade_mu
    cs = (nu - u.c) @ r
    return (sub + cs) / max(1, abs(dual)), viol / sc, dual, primal, sub, cs

nu0 = cfmm.start_prices(net, u)
s = np.log(nu0); s[tgt] = 0.0
free = np.ones(n, bool); free[tgt] = False
ex = evaluate(s, 0.0)
mu = 1e-3 * abs(ex["g"]) / NB * 100
print("pools", NB // 2, "g0", ex["g"], "mu0", mu)
t0 = time.time(); nev = 0
for it in range(200):
    e = evaluate(s, mu, True); nev += 1
    G = e["G"]; H = e["H"] + np.diag(np.maximum(G, 0))
    gap, inf, dual, primal, sub, cs = certs(s, e["psi"], e["tradeval"])
    dec = 0.0
    Hr = H[np.ix_(free, free)]
    try:
        cf = sla.cho_factor(Hr + 1e-14 * np.trace(Hr) / n * np.eye(n - 1))
        d = np.zeros(n); d[free] = -sla.cho_solve(cf, G[free])
    except Exception as ex_:
        print("chol failed", ex_); break
    dec = -(G @ d)
    print("%3d ev %3d mu %.2e g_mu %.10g dual %.10g primal %.10g gap %.2e (sub %.1e cs %.1e) infeas %.2e dec %.2e |d| %.2e"
          % (it, nev, mu, e["g"], dual, primal, gap, sub, cs, inf, dec, np.abs(d).max()))
    if abs(gap) <= 1e-6 and inf <= 1e-6:
        print("CONVERGED"); break
    t = min(1.0, 2.0 / max(np.abs(d).max(), 1e-300))
    for ls in range(40):
        e2 = evaluate(s + t * d, mu); nev += 1
        if e2["g"] <= e["g"] - 1e-4 * t * dec or (dec < 1e-13 * abs(e["g"])): break
        t *= 0.5
    s = s + t * d
    if t == 1.0 or dec < 1e-3 * mu * NB:
        mu = max(mu * sigma, 1e-16 * abs(dual) / NB * 1e3) if dec < 10 * mu * NB else mu
print("time", time.time() - t0, "evals", nev)
