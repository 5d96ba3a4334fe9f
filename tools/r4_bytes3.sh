#!/bin/bash
cd "$(dirname "$0")/.."
for wide in -1 0; do for cp in -1 0; do
  echo "== wide=$wide compact=$cp"
  CFMM_WIDE=$wide CFMM_COMPACT=$cp python - <<'PY'
import sys, json
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np, cfmm
from cfmm import synthetic, _lib
for name, scale in (("C4", 1.0), ("C4", 4.0)):
    net = synthetic.config(name, scale=scale, seed=0)
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    p.solve(tol=1e-6)
    its = []
    for _ in range(4):
        p.solve(tol=1e-6); its.append(1e6 * p.stats["device_seconds"] / p.stats["evals"])
    p.ctx.set_nu(net["c"] * np.exp(np.random.default_rng(0).normal(0, 0.01, net["n_tokens"])))
    us = min(1e6 * p.ctx.time_eval_kernel(_lib.TIME_ALL, 30) for _ in range(3))
    print("  %s x%g pools %d eval_us %.2f iter_us %.2f evals %d status %s" % (name, scale, p.m, us, min(its), p.stats["evals"], p.status))
    p.close()
PY
done; done
python tools/kernel_budget.py --only C3 C4shard C2 | tail -1
