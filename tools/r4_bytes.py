"""evaluation time of constant-product pool sets around the Infinity-Cache size (what 28 instead of 32 bytes per pool would buy)"""
import sys, json
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np, cfmm
from cfmm import synthetic, _lib
for scale in (0.5, 0.7, 0.8, 0.875, 1.0, 1.25, 3.5, 4.0):
    net = synthetic.config("C4", scale=scale, seed=0)
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    p.solve(tol=1e-6)
    it = 1e6 * p.stats["device_seconds"] / p.stats["evals"]
    p.ctx.set_nu(net["c"] * np.exp(np.random.default_rng(0).normal(0, 0.01, net["n_tokens"])))
    us = min(1e6 * p.ctx.time_eval_kernel(_lib.TIME_ALL, 30) for _ in range(3))
    m = p.m
    print(json.dumps(dict(pools=m, MB=round(32e-6 * m, 1), eval_us=round(us, 2), GBps=round(32 * m / us / 1e3, 1), iter_us=round(it, 2), ns_per_kpool=round(1e3 * us / (m / 1e3), 3))))
    p.close()
