#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
python - <<'PY'
import sys
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import numpy as np, cfmm
from cfmm import synthetic, _lib
net = synthetic.make_network(200, m_cp2=20000, m_gn=2000, m_gk_stable=1000, seed=3)
bad = 0
for rep in range(40):
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    v = p.solve(tol=1e-6, max_evals=4000, method="lbfgs")
    f, psi = p.eval_dual(p.nu.copy())
    v2 = p.solve(tol=1e-6, method="newton")
    ok = p.status == "optimal" and p.stats["newton_steps"] <= 40
    if not ok:
        bad += 1
        print(rep, p.status, p.stats["newton_steps"], p.stats["evals"], p.gap, p.infeas, p.stats.get("barrier_mu"))
    p.close()
print("failures", bad, "of 40")
PY
