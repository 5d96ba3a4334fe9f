#!/usr/bin/env python
"""Profiling target: a network of stableswap BASKETS (the K-asset table's pools) through the second-order path -- 1e5 four-asset
stableswap pools + 5e4 constant-product pools over 1000 tokens, basket liquidation; a few cold solves, one JSON line.
    rocprofv3 --kernel-trace --stats -- python tools/profile_table_newton.py [--pools 100000] [--solves 3]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    sys.path.insert(0, p)
import numpy as np
import cfmm
from cfmm import synthetic
ap = argparse.ArgumentParser()
ap.add_argument("--pools", type=int, default=100000); ap.add_argument("--assets", type=int, default=4); ap.add_argument("--solves", type=int, default=3)
ap.add_argument("--method", default="newton")
a = ap.parse_args()
net = synthetic.make_network(1000, m_cp2=50000, m_gk_stable=a.pools, gk_sizes=(a.assets, a.assets), seed=3)
rng = np.random.default_rng(5)
h = np.zeros(1000); basket = rng.choice(1000, 10, replace=False); t = int(basket[0]); h[basket[1:]] = 50.0 / net["prices"][basket[1:]]
for name, util in (("arbitrage", cfmm.Arbitrage(net["c"])), ("liquidate", cfmm.Liquidate(h, t))):
    p = cfmm.Problem.from_network(net, utility=util)
    out = []
    for rep in range(a.solves + 1):
        t0 = time.perf_counter(); v = p.solve(tol=1e-6, method=a.method, max_evals=4000); dt = time.perf_counter() - t0
        if rep:
            out.append(dict(ms=1e3 * dt, device_ms=1e3 * p.stats["device_seconds"], evals=p.stats["evals"], newton_steps=p.stats.get("newton_steps"), status=p.status, gap=p.gap, infeas=p.infeas, value=v))
    print(json.dumps(dict(workload=f"{a.pools} {a.assets}-asset stableswap pools + 50000 constant-product pools / 1000 tokens, {name}", method=a.method, solves=out)))
    p.close()
