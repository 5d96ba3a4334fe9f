"""TEST INFRASTRUCTURE (oracle side) -- an independent second-order solve of the routing program in NumPy.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

The reference hands the whole program to an interior-point solver (cp.Problem.solve(), /root/reference/arbitrage.py:81-82,
liquidation.py:83-84).  This is the CPU counterpart used to pin the HIP library's second-order path at sizes no SciPy
primal reaches (BASELINE config 5: 5.5e5 pools, liquidation.py:57,77-80): plain barrier path-following on the dual of
the decomposed program,

    minimise over log-prices s   g_mu(s) = sum_i arb_i^mu(e^s) + (e^s - c)'h  - mu sum_{GE, c = 0} s_j,

with the smoothed pool subproblems of oracle/barrier_np.py (bisection + Newton polish per pool direction), LAPACK's
Cholesky for the Newton system, a FIXED shrink of the barrier weight and a fixed number of centring steps per weight.
It shares no code and no schedule with the device (which warm-starts from first-order iterations, adapts the weight to
the Newton decrement, carries low-order log-prices and stops on its own stall rules); the two agree only if both
minimise the same dual.  Certificates come from the EXACT dual evaluation (oracle/c_oracle.py or pools_np) at the
final prices, not from the smoothed one.

PARITY UNPINNED by the reference (cvxpy is not installed here): pinned against the SciPy primal restatement
(oracle/primal_scipy.py) on small instances -- tests/test_oracle.py.
"""
import numpy as np

from oracle import barrier_np

GE, EQ, FREE = 0, 1, 2


def solve(net, c, h=None, ctype=None, nu0=None, tol=1e-7, shrink=0.2, centring=2, max_steps=200, exact_eval=None, log=None):
    """-> dict(nu, psi, dual_value, primal_value, gap, infeas, steps, evals, mu).

    exact_eval(nu) -> (sum_i arb_i(nu), psi(nu)): the unsmoothed dual evaluation used for the certificates."""
    n = net["n_tokens"]
    c = np.asarray(c, float)
    h = np.zeros(n) if h is None else np.asarray(h, float)
    ct = np.zeros(n, int) if ctype is None else np.asarray(ctype, int)
    free = ct == FREE
    bar = (ct == GE) & ~(c > 0)                     # multiplier nu_j > 0 kept inside by -mu log nu_j
    lob = np.where((ct == GE) & (c > 0), np.log(np.where(c > 0, c, 1.0)), -np.inf)      # nu_j >= c_j: projection
    if (free & ~(c > 0)).any():
        raise ValueError("an unconstrained token with c = 0: unbounded")
    nu = np.asarray(net["prices"] if nu0 is None else nu0, float).copy()
    s = np.log(nu)
    s[free] = np.log(c[free])
    s = np.maximum(s, lob)
    br = barrier_np.branches(net)
    nbar = len(br["Ri"]) + int(bar.sum())            # one barrier term per pool direction (+ the multipliers')
    evals = 0

    def smooth(s, mu, hess):
        nonlocal evals
        evals += 1
        e = barrier_np.smooth_eval(net, np.exp(s), mu, hessian=hess)
        nu = np.exp(s)
        g = e["value"] + float((nu - c) @ h) - mu * float(s[bar].sum())
        G = nu * (e["psi"] + h) - mu * bar
        G[free] = 0.0
        return g, G, e

    f0, psi0 = exact_eval(np.exp(s))
    dual = f0 + float((np.exp(s) - c) @ h)
    mu = 0.1 * max(abs(dual), 1e-300) / nbar
    steps = 0
    final = False
    while True:
        for _ in range(centring if not final else 50):
            g, G, e = smooth(s, mu, True)
            lobf = np.where(np.isfinite(lob), lob, 0.0)
            pin = free | (np.isfinite(lob) & (s <= lobf + 1e-13 * np.maximum(1.0, np.abs(lobf))) & (G > 0))
            G = np.where(pin, 0.0, G)
            H = e["H"] + np.diag(np.maximum(G, 0.0))
            H[pin, :] = 0.0; H[:, pin] = 0.0; H[pin, pin] = 1.0
            reg = 0.0
            while True:
                try:
                    Lc = np.linalg.cholesky(H + reg * np.eye(n))
                    break
                except np.linalg.LinAlgError:
                    reg = max(10.0 * reg, 1e-12 * np.abs(np.diag(H)).max())
            d = -np.linalg.solve(Lc.T, np.linalg.solve(Lc, G))
            dec = float(-G @ d)
            t = min(1.0, 2.0 / max(np.abs(d).max(), 1e-300))
            while True:
                s2 = np.maximum(s + t * d, lob)
                s2[free] = s[free]
                g2, _, _ = smooth(s2, mu, False)
                if g2 <= g + 1e-4 * float(G @ (s2 - s)) or dec <= 1e-13 * abs(g) or t < 1e-12:
                    break
                t *= 0.5
            s = s2
            steps += 1
            if log:
                log("step %d mu %.3e g %.10g dec %.3e t %.3g" % (steps, mu, g2, dec, t))
            if final:
                # certificates at the current point: exact dual value against the smoothed, pool-feasible primal point
                _, _, e = smooth(s, mu, False)
                nu = np.exp(s)
                fx, _ = exact_eval(nu)
                dual = fx + float((nu - c) @ h)
                r = e["psi"] + h
                viol = np.where(ct == GE, np.maximum(-r, 0.0), np.where(ct == EQ, np.abs(r), 0.0)).max()
                infeas = viol / max(np.abs(e["psi"]).max(), np.abs(h).max(), 1e-300)
                primal = float(c @ e["psi"])
                cs = float((nu - c) @ r)
                sub = max(fx - e["trade"], 0.0)
                gap = (sub + cs) / max(1.0, abs(dual))
                if log:
                    log("   dual %.12g primal %.12g gap %.3e infeas %.3e" % (dual, primal, gap, infeas))
                if (abs(gap) <= tol and infeas <= tol) or steps >= max_steps or dec <= 1e-14 * abs(g):
                    return dict(nu=nu, psi=e["psi"], dual_value=dual, primal_value=primal, gap=gap, infeas=infeas, steps=steps,
                                evals=evals, mu=mu)
            if steps >= max_steps:
                final = True
        if not final:
            dual = g2                                   # (the smoothed dual value at the current centre: within mu nbar of the exact one)
            if mu * nbar <= 0.25 * tol * max(1.0, abs(dual)):
                final = True
            else:
                mu *= shrink
