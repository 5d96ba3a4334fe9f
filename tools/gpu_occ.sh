#!/bin/bash
cd "$(dirname "$0")/.."
python - <<'PY'
import sys, json
sys.path[:0] = ['.', 'cfmm-routing-code_amd']
import numpy as np
import cfmm
from cfmm import synthetic, _lib
for name, kw in (("uniform", dict()), ("zipf1.1", dict(zipf_s=1.1)), ("zipf1.5", dict(zipf_s=1.5))):
    net = synthetic.make_network(1000, m_cp2=1_000_000, seed=0, **kw)
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    ctx = p._ensure_ctx()
    ctx.set_nu(net["c"] * np.exp(np.random.default_rng(0).normal(0, 0.01, 1000)))
    us = ctx.time_eval_kernel(_lib.TIME_ALL, 30) * 1e6
    cnt = np.bincount(net["cp2"]["ia"], minlength=1000)
    v = p.solve(tol=1e-6)
    print(name, "eval_us %.2f" % us, "max token share %.3f" % (cnt.max() / cnt.sum()), "solve", p.status, p.stats["evals"], "dev_us/eval %.1f" % (1e6 * p.stats["device_seconds"] / p.stats["evals"]))
    p.close()
PY
