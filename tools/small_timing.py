#!/usr/bin/env python
"""Latency of the reference's own problem sizes (arbitrage.py / liquidation.py: 5 pools, 4 tokens; two-asset.py: 5 pools,
3 tokens, 50-point sweep) and of small synthetic networks: wall ms per prob.solve(), evaluations, device ms."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import cfmm  # noqa: E402
from cfmm import synthetic  # noqa: E402
from oracle import instances as I  # noqa: E402
from helpers import problem_of  # noqa: E402

out = {}
for name, inst in (("arbitrage", I.arbitrage()), ("liquidation", I.liquidation())):
    p = problem_of(inst)
    p.solve(tol=1e-8)
    t0 = time.perf_counter(); ev = 0; dev = 0.0; lib = 0.0
    for _ in range(50):
        p.solve(tol=1e-8); ev += p.stats["evals"]; dev += p.stats["device_seconds"]; lib += p.stats["wall_seconds"]
    w = time.perf_counter() - t0
    out[name] = dict(ms_per_solve=1e3 * w / 50, library_ms=1e3 * lib / 50, device_ms=1e3 * dev / 50, evals=ev / 50, rounds=p.stats["rounds"], status=p.status, value=p.value)
    p.close()
p = problem_of(I.two_asset(0.0))
t0 = time.perf_counter(); ev = 0
for t in I.two_asset_sweep():
    p.set_utility(cfmm.Swap([t, 0, 0], 2)); p.solve(tol=1e-8, warm_start=True); ev += p.stats["evals"]
out["two_asset_sweep_50"] = dict(ms_total=1e3 * (time.perf_counter() - t0), evals=ev, how="50 prob.solve() calls, each warm-started from its neighbour")
# the same 50 points through ONE library call (cfmm_solve_sweep: one workgroup per point and round, the kink loop inside the library)
utils = [cfmm.Swap([t, 0, 0], 2) for t in I.two_asset_sweep()]
p.solve_many(utils, tol=1e-8)
best, tot = 1e9, 0.0
for _ in range(20):
    t0 = time.perf_counter(); res = p.solve_many(utils, tol=1e-8); dt = time.perf_counter() - t0
    best = min(best, dt); tot += dt
out["two_asset_sweep_50_one_call"] = dict(ms_total_best=1e3 * best, ms_total_mean=1e3 * tot / 20, library_ms=1e3 * res[0]["stats"]["wall_seconds"],
                                         device_ms=1e3 * res[0]["stats"]["device_seconds"], evals=sum(r["stats"]["evals"] for r in res),
                                         rounds_max=max(r["stats"]["rounds"] for r in res), points_on_a_kink=sum(1 for r in res if r["stats"]["rounds"] > 1),
                                         status=sorted(set(r["status"] for r in res)), how="Problem.solve_many -> cfmm_solve_sweep, cold starts, tenders of every point included")
p.close()
for m, n in ((100, 20), (1000, 50), (10000, 200), (100000, 1000)):
    net = synthetic.make_network(n, m_cp2=int(0.7 * m), m_w2=int(0.2 * m), m_gn=int(0.1 * m), seed=1)
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    p.solve(tol=1e-6)
    t0 = time.perf_counter(); ev = 0; dev = 0.0
    for _ in range(20):
        p.solve(tol=1e-6); ev += p.stats["evals"]; dev += p.stats["device_seconds"]
    w = time.perf_counter() - t0
    out[f"synthetic_{m}x{n}"] = dict(ms_per_solve=1e3 * w / 20, device_ms=1e3 * dev / 20, evals=ev / 20, us_per_eval=1e6 * dev / ev, status=p.status)
    p.close()
print(json.dumps(out))
