import sys, json, time
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np
import cfmm
from cfmm import synthetic
net = synthetic.config("C5")
n = net["n_tokens"]
rng = np.random.default_rng(1)
h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
t = int(rng.integers(0, n)); h[t] = 0
for util_name, util in (("liquidate", cfmm.Liquidate(h, t)), ("arbitrage", cfmm.Arbitrage(net["c"]))):
    p = cfmm.Problem.from_network(net, utility=util)
    for shrink in (0.1, 0.03, 0.01, 0.003):
        ctx = p._ensure_ctx(); p._send_utility()
        nu0 = cfmm.problem.start_prices(net, util)
        ctx.solve(nu0, tol=1e-6, method="newton", barrier_shrink=shrink)
        t0 = time.time(); s = ctx.solve(nu0, tol=1e-6, method="newton", barrier_shrink=shrink); wall = time.time() - t0
        class P: pass
        p.status = s["status"]; p.value = s["primal_value"]; p.gap = s["gap"]; p.infeas = s["infeas"]
        print(json.dumps(dict(util=util_name, shrink=shrink, status=p.status, value=p.value, gap=p.gap, infeas=p.infeas, steps=s["newton_steps"], evals=s["evals"],
                              dev_ms=s["device_seconds"] * 1e3, wall_ms=wall * 1e3)))
    p.close()
