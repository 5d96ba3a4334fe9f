"""where the wall time of a solve goes beyond the device time: Python around the C call (Problem.solve), the C call itself
(cfmm_solve: entry checks, its timed loop = stats.wall_seconds, exit), the device (events)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")]
import numpy as np, cfmm
from cfmm import synthetic
for name in ("C3", "C2"):
    net = synthetic.config(name, seed=0)
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    for _ in range(5): p.solve(tol=1e-6)
    raw = p.ctx.L.cfmm_solve
    tc = [0.0]
    class Timed:
        def __call__(self, *a):
            t0 = time.perf_counter(); r = raw(*a); tc[0] = time.perf_counter() - t0; return r
    class LWrap:
        def __init__(self, L): self.__dict__["_L"] = L
        def __getattr__(self, k): return Timed() if k == "cfmm_solve" else getattr(self._L, k)
    p.ctx.L = LWrap(p.ctx.L)
    rows = []
    for _ in range(30):
        t0 = time.perf_counter(); p.solve(tol=1e-6); t1 = time.perf_counter()
        rows.append((1e6 * (t1 - t0), 1e6 * tc[0], 1e6 * p.stats["wall_seconds"], 1e6 * p.stats["device_seconds"], p.stats["evals"]))
    a = np.median(np.array(rows), axis=0)
    print("%s: Problem.solve %.1f us | cfmm_solve call %.1f | its timed loop %.1f | device %.1f | evals %d  => python %.1f, C entry/exit %.1f, loop beyond device %.1f"
          % (name, a[0], a[1], a[2], a[3], a[4], a[0] - a[1], a[1] - a[2], a[2] - a[3]))
    p.close()
