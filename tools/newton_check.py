"""GPU check of the second-order path: smoothed evaluation vs the NumPy oracle, then solves of config 5."""
import sys, time, json
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np
import cfmm
from cfmm import synthetic
from oracle import barrier_np


def basket(net, seed=1, k=10):
    n = net["n_tokens"]
    rng = np.random.default_rng(seed)
    h = np.zeros(n); idx = rng.choice(n, k, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, k)) / net["prices"][idx] * 10
    tgt = int(rng.integers(0, n)); h[tgt] = 0
    return h, tgt


def main():
    scales = [float(x) for x in sys.argv[1:]] or [0.02]
    net = synthetic.config("C5", scale=0.02)
    h, tgt = basket(net)
    u = cfmm.Liquidate(h, tgt)
    p = cfmm.Problem.from_network(net, utility=u)
    nu0 = cfmm.start_prices(net, u)
    ctx = p._ensure_ctx()
    rng = np.random.default_rng(0)
    M = rng.normal(size=(net["n_tokens"], 40)); A = M @ M.T + np.diag(rng.uniform(0.5, 2.0, net["n_tokens"])); b = rng.normal(size=net["n_tokens"])
    x, info = ctx.debug_cholesky(A, b)
    t = time.time(); x, info = ctx.debug_cholesky(A, b); t = time.time() - t
    xr = np.linalg.solve(A, b)
    print("cholesky n=%d: info %d, rel err %.2e, residual %.2e, %.2f ms incl. 8 MB upload" % (len(b), info, np.abs(x - xr).max() / np.abs(xr).max(), np.abs(A @ x - b).max(), t * 1e3))
    x, info = ctx.debug_cholesky(A - 3 * np.eye(len(b)), b)
    print("   indefinite matrix -> info", info)
    for mu in (1e-2, 1e-6, 1e-11):
        t = time.time(); val, tr, psi, H = ctx.eval_smooth(nu0, mu, want_hessian=True); t = time.time() - t
        o = barrier_np.smooth_eval(net, nu0, mu, hessian=True)
        Hl = np.tril(H); Ho = np.tril(o["H"])
        print("mu %.0e: value %.12g vs %.12g | trade %.12g vs %.12g | psi err %.2e (scale %.2e) | H err %.2e (scale %.2e) | %.1f ms"
              % (mu, val, o["value"], tr, o["trade"], np.abs(psi - o["psi"]).max(), np.abs(o["psi"]).max(),
                 np.abs(Hl - Ho).max(), np.abs(Ho).max(), t * 1e3))
    for sc in scales:
        net = synthetic.config("C5", scale=sc)
        h, tgt = basket(net)
        u = cfmm.Liquidate(h, tgt)
        p = cfmm.Problem.from_network(net, utility=u)
        for rep in range(2):
            t = time.time(); p.solve(method="newton"); t = time.time() - t
            s = p.stats
            print("C5 scale %g: %s value %.10g dual %.10g gap %.2e infeas %.2e | newton %d evals %d mu %.1e | wall %.1f ms device %.1f ms (host %.1f ms)"
                  % (sc, p.status, p.value, p.dual_value, p.gap, p.infeas, s["newton_steps"], s["evals"], s["barrier_mu"],
                     s["wall_seconds"] * 1e3, s["device_seconds"] * 1e3, t * 1e3))
        d, l = p.bucket_trades("curve2")
        print("   curve2 tenders: min delta %.2e, pools with delta > 1e-9 R: %d of %d" % (d.min(), int((d > 1e-9 * net["curve2"]["Ra"]).sum()), d.size))


main()
