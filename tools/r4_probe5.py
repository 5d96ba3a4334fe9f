import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cfmm-routing-code_amd"))
import numpy as np, cfmm
from cfmm import synthetic
for ms, mk in ((4000, 0), (4000, 50), (1000, 0), (0, 1000)):
    net = synthetic.make_network(200, m_cp2=20000, m_gn=2000, m_gk_stable=ms, m_gk_sum=mk, seed=3)
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    t0 = time.perf_counter(); v = p.solve(tol=1e-6, max_evals=4000); dt = time.perf_counter() - t0
    print("stable", ms, "sum", mk, p.status, "evals", p.stats["evals"], "gap %.2e infeas %.2e" % (p.gap, p.infeas), "ms %.1f" % (1e3 * dt), flush=True)
    p.close()
