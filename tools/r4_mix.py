"""evaluation cost per pool kind AT SCALE: networks of one kind each, a full chip's worth of pools (tools/README.md)"""
import sys, json
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np
import cfmm
from cfmm import synthetic, _lib
cases = {
    "C3": dict(m_cp2=700_000, m_w2=200_000, m_gn=100_000),
    "C3 without K-asset": dict(m_cp2=700_000, m_w2=200_000),
    "cp2 1e6": dict(m_cp2=1_000_000),
    "w2 1e6": dict(m_w2=1_000_000),
    "gn3 1e6": dict(m_gn=1_000_000, gn_sizes=(3, 3)),
    "gn5 5e5": dict(m_gn=500_000, gn_sizes=(5, 5)),
    "gn8 5e5": dict(m_gn=500_000, gn_sizes=(8, 8)),
    "gn3-8 5e5": dict(m_gn=500_000),
    "K-asset 1e5 alone": dict(m_gn=100_000),
}
for name, kw in cases.items():
    net = synthetic.make_network(1000, seed=0, **kw)
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    p.solve(tol=1e-6)
    p.ctx.set_nu(net["c"] * np.exp(np.random.default_rng(0).normal(0, 0.01, net["n_tokens"])))
    us = min(1e6 * p.ctx.time_eval_kernel(_lib.TIME_ALL, 50) for _ in range(3))
    print(json.dumps(dict(case=name, pools=int(p.m), eval_us=round(us, 2), iter_us=round(1e6 * p.stats["device_seconds"] / p.stats["evals"], 2), evals=p.stats["evals"])))
    p.close()
