#!/bin/bash
cd "$(dirname "$0")/.."
python - <<'PY'
import sys, time
sys.path[:0] = ['.', 'cfmm-routing-code_amd']
import numpy as np
import cfmm
from cfmm import synthetic
net = synthetic.config("C3", seed=0)
for rep in range(3):
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    t0 = time.perf_counter(); p._ensure_ctx(); t1 = time.perf_counter()
    v = p.solve(tol=1e-6); t2 = time.perf_counter()
    d = p.bucket_trades("cp2"); t3 = time.perf_counter()
    print("create+upload %.2f ms  first solve %.2f ms  trades(cp2) %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    p.close()
PY
