"""Config 5, second-order path: the per-step trace (CFMM_NEWTON_TRACE) of one liquidation solve."""
import os, sys, time
os.environ["CFMM_NEWTON_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")]
import numpy as np, cfmm, bench
from cfmm import synthetic
net = synthetic.config("C5")
prob = bench.make_problem(net, "C5") if hasattr(bench, "make_problem") else None
if prob is None:
    rng = np.random.default_rng(1); n = net["n_tokens"]
    h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
    t = int(rng.integers(0, n)); h[t] = 0
    prob = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t))
prob.solve()
t0 = time.time(); prob.solve(); print("solve ms", (time.time() - t0) * 1e3, prob.stats)
