"""second-order prelude length on config 5's network under BOTH utilities of round 3's sweep (liquidation of a basket; linear arbitrage)"""
import sys, os, json, subprocess
if len(sys.argv) == 1:
    for pre in (4, 6, 8, 10, 12, 16):
        env = dict(os.environ, CFMM_NEWTON_PRELUDE=str(pre))
        print(subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True).stdout.strip())
    sys.exit(0)
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np, cfmm
from cfmm import synthetic
net = synthetic.config("C5"); n = net["n_tokens"]; rng = np.random.default_rng(1)
h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
t = int(rng.integers(0, n)); h[t] = 0
out = {"prelude": os.environ.get("CFMM_NEWTON_PRELUDE")}
for name, u in (("liquidate", cfmm.Liquidate(h, t)), ("arbitrage", cfmm.Arbitrage(net["c"]))):
    p = cfmm.Problem.from_network(net, utility=u)
    ms = []
    for _ in range(6):
        p.solve(method="newton"); ms.append(p.stats["wall_seconds"] * 1e3)
    out[name] = dict(ms=round(sorted(ms)[2], 3), steps=p.stats["newton_steps"], evals=p.stats["evals"], status=p.status)
    p.close()
print(json.dumps(out))
