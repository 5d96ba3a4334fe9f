#!/usr/bin/env python
"""Profiling target / bench line of the UTILITY TABLE's first-order path: 5e4 pools of every reference kind / 1000
tokens under a logarithmic utility (CFMM_ULOG on every token; the generic two-launch iteration, L-BFGS memory 8) and the same
instance through the second-order path: one JSON line -- evaluations, ms per solve, device us per evaluation, pool-subproblems/s."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    sys.path.insert(0, p)
import numpy as np
import cfmm
from cfmm import synthetic
net = synthetic.config("C3", scale=0.05, seed=2)
n = net["n_tokens"]
rng = np.random.default_rng(0)
hold = np.exp(rng.normal(3, 0.5, n)) / net["prices"]
u = cfmm.LogUtility(hold * net["prices"] * np.exp(rng.normal(0, 0.1, n)), hold)
p = cfmm.Problem.from_network(net, utility=u)
out = dict(workload=f"{p.m} pools / {n} tokens, u = sum a_k log(psi_k + h_k) on every token (weights within 10 % of the market)")
for method in ("lbfgs", "newton"):
    p.solve(tol=1e-6, method=method)
    t0 = time.perf_counter(); ev = dev = 0
    for _ in range(5):
        v = p.solve(tol=1e-6, method=method); ev += p.stats["evals"]; dev += p.stats["device_seconds"]
    dt = time.perf_counter() - t0
    out[method] = dict(status=p.status, value=v, gap=p.gap, infeas=p.infeas, evals_per_solve=ev / 5, ms_per_solve=1e3 * dt / 5,
                       device_us_per_evaluation=1e6 * dev / ev, pool_subproblems_per_s=ev * p.m / dt, newton_steps=p.stats.get("newton_steps", 0))
print(json.dumps(out))
p.close()
