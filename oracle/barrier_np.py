"""TEST INFRASTRUCTURE (oracle side) -- NumPy restatement of the barrier-smoothed pool subproblems.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

The reference solves the whole routing program with an interior-point method (cp.Problem.solve(),
/root/reference/arbitrage.py:81-82).  The second-order path of the HIP library puts the same log barrier on
Delta, Lambda >= 0 (arbitrage.py:51-52) inside the dual decomposition; this module restates that smoothed
evaluation on the CPU, with a deliberately different inner solver (bisection in log D followed by plain
Newton polish, instead of the device's safeguarded barrier-exact iteration), so that the two only agree if
both solve the same one-dimensional problem

    max_{D > 0}  nu_out L(D) - nu_in D + mu log D          (constant sum: + mu log(R_out/gamma - D))

per pool direction, L = forward exchange function of the pool (arbitrage.py:60,63-74).  k-asset geo-mean pools
(arbitrage.py:65) enter unsmoothed, with their exact solution and generalised Hessian.

PARITY UNPINNED by the reference (it has no smoothed evaluation to compare with): the smoothed quantities are
pinned against finite differences of themselves and, through the solves they drive, against the SciPy primal
(oracle/primal_scipy.py) -- see tests/test_gpu.py.
"""
import numpy as np

KCP, KW, KSUM, KCV, KPW = 0, 1, 2, 3, 4          # CFMM_POOL_* codes


def branches(net):
    """every two-asset pool as two directed branches (tender `ti`, receive `to`)"""
    cols = dict(Ri=[], Ro=[], fee=[], par=[], ti=[], to=[], kind=[], pool=[], key=[])

    def add(kind, key, b, pa=None, pb=None):
        m = len(b["Ra"]); z = np.zeros(m)
        cols["Ri"] += [b["Ra"], b["Rb"]]; cols["Ro"] += [b["Rb"], b["Ra"]]; cols["fee"] += [b["fee"]] * 2
        cols["par"] += [z if pa is None else pa, z if pb is None else pb]
        cols["ti"] += [b["ia"], b["ib"]]; cols["to"] += [b["ib"], b["ia"]]
        cols["kind"] += [np.full(2 * m, kind)]
    if "cp2" in net: add(KCP, "cp2", net["cp2"])
    if "w2" in net:
        b = net["w2"]; add(KW, "w2", b, b["wa"] / (1 - b["wa"]), (1 - b["wa"]) / b["wa"])
    if "curve2" in net:
        b = net["curve2"]; add(KCV, "curve2", b, b["alpha"], b["alpha"])
    if "pow2" in net:
        b = net["pow2"]; add(KPW, "pow2", b, b["t"], b["t"])
    if "sum2" in net: add(KSUM, "sum2", net["sum2"])
    out = {k: np.concatenate(v) for k, v in cols.items() if v}
    out["Ri"] = out["Ri"].astype(float); out["Ro"] = out["Ro"].astype(float)
    return out


def forward(br, D):
    """L(D), L'(D), L''(D) per branch"""
    Ri, Ro, fee, par, kind = br["Ri"], br["Ro"], br["fee"], br["par"], br["kind"]
    x = Ri + fee * D
    kp = Ri * Ro
    with np.errstate(all="ignore"):
        Lp = fee * D * Ro / x; L1p = fee * kp / (x * x); L2p = -2 * fee * fee * kp / x ** 3
        r = par
        lq = -r * np.log1p(fee * D / Ri)
        Lw = -Ro * np.expm1(lq); L1w = fee * Ro * r * np.exp(lq) / x; L2w = -fee * (r + 1) * L1w / x
        al = par
        Kc = Ri + Ro - al / kp
        b = Kc - x; q = 4 * al / x; sq = np.sqrt(b * b + q)
        Y = np.where(b >= 0, 0.5 * (b + sq), 0.5 * q / (sq - b))
        fx = 1 + al / (x * x * Y); fy = 1 + al / (x * Y * Y); Y1 = -fx / fy
        fxx = -2 * al / (x ** 3 * Y); fxy = -al / (x * x * Y * Y); fyy = -2 * al / (x * Y ** 3)
        Y2 = -(fxx + 2 * fxy * Y1 + fyy * Y1 * Y1) / fy
        Lc = Ro - Y; L1c = -fee * Y1; L2c = -fee * fee * Y2
        # power sum x^q + y^q, q = 1 - t (par = t): y from the level set directly (a different route from the library's
        # log1p / expm1 form), L' = gamma (y / x)^t, L'' = -t L' (L' / y + gamma / x)
        tq = np.where(kind == KPW, par, 0.5); qq = 1.0 - tq
        yq = np.maximum(Ro ** qq - (x ** qq - Ri ** qq), 0.0)
        Yp = yq ** (1.0 / qq)
        Lpw = Ro - Yp; L1pw = np.where(Yp > 0, fee * (Yp / x) ** tq, 0.0)
        L2pw = np.where(Yp > 0, -tq * L1pw * (L1pw / np.where(Yp > 0, Yp, 1.0) + fee / x), -1e-300)
    ks = [kind == KCP, kind == KW, kind == KCV, kind == KPW]
    L = np.select(ks, [Lp, Lw, Lc, Lpw], fee * D)
    L1 = np.select(ks, [L1p, L1w, L1c, L1pw], fee)
    L2 = np.select(ks, [L2p, L2w, L2c, L2pw], 0.0)
    return L, L1, L2


def solve_branches(br, nu, mu):
    """D*, L(D*), kappa, L'(D*), value per branch"""
    ni, no = nu[br["ti"]], nu[br["to"]]
    issum = br["kind"] == KSUM
    cap = np.where(issum, br["Ro"] / br["fee"], np.inf)

    def F(D):
        L, L1, L2 = forward(br, D)
        with np.errstate(all="ignore"):
            f = no * L1 - ni + mu / D - np.where(issum, mu / (cap - D), 0.0)
            fd = no * L2 - mu / (D * D) - np.where(issum, mu / (cap - D) ** 2, 0.0)
        return f, fd
    # F is decreasing in D: bisection on u = log D (constant sum: on D/cap in (0, 1) through a logit)
    lo = np.full(len(ni), -90.0); hi = np.full(len(ni), 60.0)
    scale = br["Ri"]

    def point(u):
        with np.errstate(all="ignore"):
            return np.where(issum, cap / (1.0 + np.exp(-u)), scale * np.exp(u))
    for _ in range(90):
        mid = 0.5 * (lo + hi)
        f, _ = F(point(mid))
        pos = f > 0
        lo = np.where(pos, mid, lo); hi = np.where(pos, hi, mid)
    D = point(0.5 * (lo + hi))
    for _ in range(3):                       # Newton polish (kept only where it stays positive and improves)
        f, fd = F(D)
        Dn = D - f / fd
        fn, _ = F(Dn)
        good = (Dn > 0) & (Dn < cap) & (np.abs(fn) < np.abs(f))
        D = np.where(good, Dn, D)
    L, L1, L2 = forward(br, D)
    _, fd = F(D)
    kappa = -1.0 / fd
    val = no * L - ni * D + mu * np.log(D) + np.where(issum, mu * np.log(np.where(issum, cap - D, 1.0)), 0.0)
    return D, L, kappa, L1, val


def smooth_eval(net, nu, mu, hessian=False, flags=None):
    """value = sum of branch optima, trade = sum nu'(L - D), psi_mu; optionally the full symmetric Hessian of
    `value` in log-prices without its diag(nu * psi) term (what cfmm_eval_smooth returns in its lower triangle)"""
    n = net["n_tokens"]
    nu = np.asarray(nu, float)
    br = branches(net)
    D, L, kappa, L1, val = solve_branches(br, nu, mu)
    ti, to = br["ti"], br["to"]
    psi = np.bincount(to, L, n) - np.bincount(ti, D, n)
    trade = float((nu[to] * L - nu[ti] * D).sum())
    out = dict(value=float(val.sum()), trade=trade, psi=psi, D=D, L=L, branches=br)
    H = np.zeros((n, n)) if hessian else None
    if hessian:
        wi = nu[ti]; wo = -L1 * nu[to]
        np.add.at(H, (ti, ti), kappa * wi * wi); np.add.at(H, (to, to), kappa * wo * wo)
        np.add.at(H, (ti, to), kappa * wi * wo); np.add.at(H, (to, ti), kappa * wi * wo)
    # k-asset geo-mean pools (arbitrage.py:65) are not smoothed: exact solution, exact generalised Hessian.  With A
    # the legs that trade, p_j c_j x_j = w_j m on either side of the fee (m = the KKT multiplier), so the block is
    # m (diag(w_A) - w_A w_A' / sum w_A); m is recovered here from the traded legs themselves.
    from oracle import pools_np
    for k, b in net.get("gn", {}).items():
        for i in range(b["R"].shape[1]):
            tok = b["idx"][:, i]; R = b["R"][:, i]; w = b["w"][:, i]; g = b["fee"][i]; p = nu[tok]
            y, v = pools_np.arb_geomean_n(R, w, g, p)
            np.add.at(out["psi"], tok, y)
            out["value"] += v; out["trade"] += v
            act = y != 0
            if hessian and act.any():
                x = np.where(y > 0, R - y, R - g * y)
                m = np.mean((p * x / np.where(y > 0, 1.0, g) / w)[act])
                wa = np.where(act, w, 0.0)
                blk = m * (np.diag(wa) - np.outer(wa, wa) / wa.sum())
                H[np.ix_(tok, tok)] += blk
    if hessian:
        out["H"] = H
    return out
