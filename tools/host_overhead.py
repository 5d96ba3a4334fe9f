#!/usr/bin/env python
"""Where the host's share of a C3 solve goes: cProfile over repeated Problem.solve() calls, plus the library's own wall /
device clocks (tuning aid; not part of bench.py's contract)."""
import cProfile, pstats, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import cfmm
from cfmm import synthetic
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
net = synthetic.config(cfg, seed=0)
prob = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
for _ in range(5):
    prob.solve(tol=1e-6)
N = 200
t0 = time.perf_counter(); wall = dev = ev = 0
for _ in range(N):
    prob.solve(tol=1e-6); wall += prob.stats["wall_seconds"]; dev += prob.stats["device_seconds"]; ev += prob.stats["evals"]
dt = time.perf_counter() - t0
print(json.dumps(dict(config=cfg, us_per_solve=1e6 * dt / N, c_wall_us=1e6 * wall / N, device_us=1e6 * dev / N, evals=ev / N,
                      python_us=1e6 * (dt - wall) / N, c_host_us=1e6 * (wall - dev) / N, run_ahead=os.environ.get("CFMM_RUN_AHEAD", "default"))))
pr = cProfile.Profile(); pr.enable()
for _ in range(N):
    prob.solve(tol=1e-6)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
