import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")): sys.path.insert(0, p)
import numpy as np, cfmm
from cfmm import synthetic
net = synthetic.make_network(200, m_cp2=20000, m_gn=2000, m_gk_stable=1000, seed=3)
p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
for rep in range(3):
    t0 = time.perf_counter(); v = p.solve(tol=1e-6, max_evals=4000, method="lbfgs"); dt = time.perf_counter() - t0
    print(os.environ.get("CFMM_TABLE_WARM"), os.environ.get("CFMM_TABLE_FTOL"), os.environ.get("CFMM_TABLE_GRID_MULT"), p.status, p.stats["evals"], "%.6f" % v, "gap %.2e infeas %.2e" % (p.gap, p.infeas), "%.1f ms" % (1e3 * dt))
t0 = time.perf_counter(); v2 = p.solve(tol=1e-6, method="newton"); dt = time.perf_counter() - t0
print("newton", p.status, p.stats["newton_steps"], p.stats["evals"], "%.6f" % v2, "gap %.2e infeas %.2e" % (p.gap, p.infeas), "%.1f ms" % (1e3 * dt))
