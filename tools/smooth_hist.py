"""iterations of smooth_kernel's per-direction root searches over a config-5 solve, and the SIMT efficiency of their loop (a -DCFMM_SMOOTH_HIST
variant: make variant TAG=hist DEFS=-DCFMM_SMOOTH_HIST; CFMM_LIB=<that library> python tools/smooth_hist.py)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")]
import numpy as np, cfmm
from cfmm import synthetic
net = synthetic.config("C5")
n = net["n_tokens"]; rng = np.random.default_rng(1)
h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
t = int(rng.integers(0, n)); h[t] = 0
p = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t))
L = p._ensure_ctx().L
L.cfmm_debug_smooth_hist.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
out = (C.c_uint64 * 128)()
p.solve(method="newton")
L.cfmm_debug_smooth_hist(p.ctx.h, out, 1)          # (first solve: cold warm-start columns) discard
p.solve(method="newton")
L.cfmm_debug_smooth_hist(p.ctx.h, out, 1)
hist = np.array(out[:126], dtype=np.int64)
tot = hist.sum()
print("direction solves", int(tot), "mean iterations", float((hist * (np.arange(126) + 1)).sum() / max(tot, 1)))
print("share by iterations (1-based):", {int(k + 1): round(float(v) / tot, 4) for k, v in enumerate(hist) if v > 1e-4 * tot})
print("lane-iterations", int(out[126]), "64 x wave maxima", int(out[127]), "SIMT efficiency of the loop", out[126] / max(out[127], 1))
