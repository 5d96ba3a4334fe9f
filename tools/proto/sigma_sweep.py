import sys
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd', '/root/repo/tests']
import numpy as np, cfmm
from cfmm import synthetic
from test_gpu_newton import _mixed_network, _basket
cases = []
net = synthetic.config("C5"); h, t = _basket(net); cases.append(("C5 liq", net, cfmm.Liquidate(h, t)))
net = synthetic.config("C5", scale=0.1); h, t = _basket(net); cases.append(("C5x0.1 swap", net, cfmm.Swap(h, t)))
net = _mixed_network(seed=3); h, t = _basket(net); cases.append(("mixed liq", net, cfmm.Liquidate(h, t)))
net = synthetic.config("C2"); cases.append(("C2 arb", net, cfmm.Arbitrage(net["c"])))
for name, net, u in cases:
    p = cfmm.Problem.from_network(net, utility=u)
    ctx = p._ensure_ctx(); ctx.set_utility(u.c, u.h, u.ctype)
    nu0 = cfmm.start_prices(net, u)
    for sg in ((0.2,) if len(sys.argv) > 1 else (0.01, 0.03, 0.05, 0.1, 0.2)):
        st = ctx.solve(nu0, method="newton", barrier_shrink=sg)
        print("%-12s sigma %.2f: status %d steps %3d evals %3d gap %.1e infeas %.1e  %.1f ms" % (name, sg, st["status"], st["newton_steps"], st["evals"], st["gap"], st["infeas"], st["wall_seconds"] * 1e3))
    p.close()
