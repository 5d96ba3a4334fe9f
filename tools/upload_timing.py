#!/usr/bin/env python
"""Host-buffer hand-over (DESIGN (g)): time cfmm_create + the uploads of a config's pool columns from pageable NumPy
buffers, the first solve on the fresh context, a later solve, and the read-back of all constant-product tenders.
Prints one JSON line; PCIe-inclusive pool-subproblems/s = pools x evals / (upload + solve)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import cfmm  # noqa: E402
from cfmm import synthetic  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
net = synthetic.config(args.config, seed=0)
nbytes = sum(len(net[k]["Ra"]) * bench.BYTES_PER_POOL[k] for k in ("cp2", "w2", "sum2", "curve2") if k in net)
nbytes += sum(b["R"].shape[1] * (20 + 20 * k) for k, b in net.get("gn", {}).items())
warm = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"])); warm._ensure_ctx(); warm.close()   # HIP runtime init
up, first, later, rb = [], [], [], []
for _ in range(args.reps):
    t0 = time.perf_counter()
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    p._ensure_ctx()
    t1 = time.perf_counter()
    p.solve(tol=1e-6)
    t2 = time.perf_counter()
    p.solve(tol=1e-6)
    t3 = time.perf_counter()
    p.bucket_trades("cp2")
    t4 = time.perf_counter()
    up.append(t1 - t0); first.append(t2 - t1); later.append(t3 - t2); rb.append(t4 - t3)
    evals = p.stats["evals"]
    p.close()
med = lambda v: sorted(v)[len(v) // 2]
print(json.dumps(dict(config=args.config, pools=p.m, column_bytes=nbytes, upload_ms=1e3 * med(up), upload_GBps=nbytes / med(up) / 1e9,
                      first_solve_ms=1e3 * med(first), later_solve_ms=1e3 * med(later), readback_cp2_ms=1e3 * med(rb),
                      evals=evals, pcie_inclusive_subproblems_per_s=p.m * evals / (med(up) + med(first)))))
