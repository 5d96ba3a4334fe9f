import sys, os, ctypes as C
# phase stamps of one workgroup of chol_step2_kernel (the launch at c0 = 512 of config 5's factorisation); needs a variant built with
#   make variant TAG=ch2stamps DEFS="-DCFMM_CH2_STAMPS -DCFMM_CH2_STAMP_WG=1"     (workgroup 1: a row workgroup; 0: the one that writes the factor)
# and CFMM_LIB pointing at it
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'cfmm-routing-code_amd')]
import numpy as np
import cfmm
from cfmm import synthetic
net = synthetic.config("C5")
n = net["n_tokens"]; rng = np.random.default_rng(1)
h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
t = int(rng.integers(0, n)); h[t] = 0
p = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t))
for _ in range(3): p.solve(method="newton")
out = (C.c_uint64 * 64)()
L = p.ctx.L
L.cfmm_debug_ch2_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
L.cfmm_debug_ch2_stamps(p.ctx.h, out)
s = np.array(out[:], dtype=np.int64).reshape(4, 16)
t0 = s[0, 0]
names = ["start", "loaded", "h0 pre done", "h0 work done", "h0 barrier", "h1 pre done", "h1 work done", "h1 barrier", "end"]
for w in range(4):
    print("wave", w, " ".join("%s=%.2f" % (names[i], (s[w, i] - t0) / 100.0) for i in range(9)))
# side-role workgroups (CFMM_CH2_STAMP_WG >= the launch's panel workgroups): kernel entry, then start / end of each task it takes
side = s[0, 9:16]
if side[0] > 0:
    print("side role: entry 0.00 " + " ".join("%s=%.2f" % (("task%d start" % ((i - 1) // 2)) if i % 2 else ("task%d end" % ((i - 2) // 2)), (side[i] - side[0]) / 100.0) for i in range(1, 7) if side[i] > side[0]))
