"""Config 5 (5e5 stableswap + 5e4 constant-product pools, 1000 tokens, basket liquidation) through the
second-order path: a few cold solves, one JSON line.  Run under rocprofv3 --kernel-trace --stats for the per-kernel
split (tools/README.md)."""
import sys, json, time, argparse
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np
import cfmm
from cfmm import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--solves", type=int, default=3)
ap.add_argument("--lbfgs-evals", type=int, default=0, help="also run the first-order method with this evaluation budget")
a = ap.parse_args()
net = synthetic.config("C5", scale=a.scale)
n = net["n_tokens"]
rng = np.random.default_rng(1)
h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
t = int(rng.integers(0, n)); h[t] = 0
p = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t))
out = dict(config="C5", scale=a.scale, pools=int(p.m), tokens=n, solves=[])
for _ in range(a.solves):
    t0 = time.time(); p.solve(method="newton"); wall = time.time() - t0
    s = p.stats
    out["solves"].append(dict(status=p.status, value=p.value, gap=p.gap, infeas=p.infeas, newton_steps=s["newton_steps"], evals=s["evals"],
                              barrier_mu=s["barrier_mu"], solve_ms=s["wall_seconds"] * 1e3, host_ms=wall * 1e3,
                              pool_subproblems_per_s=s["evals"] * p.m / s["wall_seconds"]))
if a.lbfgs_evals:
    t0 = time.time(); p.solve(method="lbfgs", max_evals=a.lbfgs_evals); wall = time.time() - t0
    out["lbfgs"] = dict(status=p.status, value=p.value, gap=p.gap, infeas=p.infeas, evals=p.stats["evals"], solve_ms=p.stats["wall_seconds"] * 1e3)
print(json.dumps(out))
