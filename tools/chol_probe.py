import sys, json
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np, cfmm
from cfmm import synthetic
net = synthetic.config("C5", scale=0.2)
p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
ctx = p._ensure_ctx(); ctx.set_nu(net["prices"])
print({k: round(v * 1e6, 1) for k, v in ctx.time_newton_kernels(1e-6, 5).items()})
