#!/bin/bash
cd "$(dirname "$0")/.."
for cfg in C3 C2 C4shard; do for ra in 1 2 3 4; do
CFMM_RUN_AHEAD=$ra python - <<PY
import sys, time, json
sys.path[:0] = ['/root/repo', '/root/repo/cfmm-routing-code_amd']
import cfmm
from cfmm import synthetic
net = synthetic.config("$cfg", seed=0)
p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
for _ in range(5): p.solve(tol=1e-6)
best = 1e9
for r in range(5):
    t0 = time.perf_counter()
    for _ in range(20): p.solve(tol=1e-6)
    best = min(best, (time.perf_counter() - t0) / 20)
print("$cfg run_ahead $ra  ms_per_solve %.4f  device_ms %.4f evals %d" % (best * 1e3, p.stats["device_seconds"] * 1e3, p.stats["evals"]))
PY
done; done
