"""one single-kind network, 60 evaluation launches (rocprofv3 target: tools/r4_mix_pmc.sh)"""
import sys
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np
import cfmm
from cfmm import synthetic, _lib
kind = sys.argv[1]
kw = {"gn": dict(m_gn=500_000), "gn3": dict(m_gn=1_000_000, gn_sizes=(3, 3)), "gn8": dict(m_gn=500_000, gn_sizes=(8, 8)), "cp2": dict(m_cp2=1_000_000), "w2": dict(m_w2=1_000_000)}[kind]
net = synthetic.make_network(1000, seed=0, **kw)
p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
p._ensure_ctx(); p._send_utility()
p.ctx.set_nu(net["c"] * np.exp(np.random.default_rng(0).normal(0, 0.01, net["n_tokens"])))
print(kind, 1e6 * p.ctx.time_eval_kernel(_lib.TIME_ALL, 60))
