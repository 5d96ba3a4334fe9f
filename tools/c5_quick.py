import sys, time, os
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np, cfmm
from cfmm import synthetic
net = synthetic.config("C5")
rng = np.random.default_rng(1); n = net["n_tokens"]
h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
t = int(rng.integers(0, n)); h[t] = 0
q = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t))
q.solve()
ts = []
for _ in range(5):
    t0 = time.time(); q.solve(); ts.append((time.time() - t0) * 1e3)
print(os.environ.get("CFMM_NEWTON_PRELUDE"), os.environ.get("CFMM_NEWTON_MU0"), "solve ms", round(min(ts), 2), round(q.stats["device_seconds"] * 1e3, 2), q.stats["newton_steps"], q.stats["evals"], q.status, q.value)
# linear arbitrage too
p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
p.solve(); t0 = time.time(); p.solve(); print("   arb ms", round((time.time() - t0) * 1e3, 2), p.stats["newton_steps"], p.stats["evals"], p.status)
