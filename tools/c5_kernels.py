import sys, json
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np, cfmm, bench
from cfmm import synthetic, _lib
net = synthetic.config("C5")
p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
p._ensure_ctx().set_nu(net["c"] * np.exp(np.random.default_rng(0).normal(0, 0.01, net["n_tokens"])))
rows = bench.kernel_table(p, 20)
print({r["kernel"]: round(r["us"], 1) for r in rows})
import time
rng = np.random.default_rng(1); n = net["n_tokens"]
h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
t = int(rng.integers(0, n)); h[t] = 0
q = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t))
q.solve(); t0 = time.time(); q.solve(); print("solve ms", (time.time() - t0) * 1e3, q.stats["device_seconds"] * 1e3, q.stats["newton_steps"], q.stats["evals"], q.status, q.value)
