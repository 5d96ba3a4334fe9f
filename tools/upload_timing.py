#!/usr/bin/env python
"""Host-buffer hand-over (DESIGN (g)): where the time of a cold `Problem` goes -- cfmm_create, the column uploads from
pageable NumPy buffers (per bucket), the first solve on the fresh context, a later solve, the read-back of the
constant-product tenders.  One JSON line; PCIe-inclusive pool-subproblems/s = pools x evals / (create + upload + solve)."""
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
from cfmm import synthetic, _lib  # noqa: E402
from cfmm.problem import KIND2  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--reps", type=int, default=7)
args = ap.parse_args()
net = synthetic.config(args.config, seed=0)
nbytes = sum(len(net[k]["Ra"]) * bench.BYTES_PER_POOL[k] for k in ("cp2", "w2", "sum2", "curve2") if k in net)
nbytes += sum(b["R"].shape[1] * (20 + 20 * k) for k, b in net.get("gn", {}).items())
warm = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"])); warm._ensure_ctx(); warm.solve(); warm.close()   # runtime init
rows = []
for _ in range(args.reps):
    t = [time.perf_counter()]
    ctx = _lib.Context(net["n_tokens"], 0); t.append(time.perf_counter())
    per = {}
    for key, kind in KIND2.items():
        if key in net:
            b = net[key]
            t0 = time.perf_counter()
            ctx.upload_pools2(kind, b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"], b.get("wa") if key == "w2" else b.get("alpha"))
            per[key] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for k, b in net.get("gn", {}).items():
        ctx.upload_poolsN(b["idx"], b["R"], b["w"], b["fee"])
    per["gn"] = time.perf_counter() - t0
    t.append(time.perf_counter())
    # the first solve through the C-ABI as INTEGRATION.md's stub drives it (utility, solve, read the prices and psi back)
    u = cfmm.Arbitrage(net["c"])
    ctx.set_utility(u.c, u.h, u.ctype); st = ctx.solve(net["c"], tol=1e-6); ctx.get_solution(); t.append(time.perf_counter())
    p = cfmm.Problem.from_network(net, utility=u); p.ctx = ctx; p._uploaded = True
    p.solve(tol=1e-6); t.append(time.perf_counter())           # the same through cfmm.Problem (host-side certificates, bookkeeping)
    p.bucket_trades("cp2"); t.append(time.perf_counter())
    rows.append(dict(create=t[1] - t[0], upload=t[2] - t[1], first=t[3] - t[2], later=t[4] - t[3], readback=t[5] - t[4], evals=st["evals"], **{"up_" + k: v for k, v in per.items()}))
    p.close()
med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]
out = dict(config=args.config, pools=cfmm.problem.network_pool_count(net), column_bytes=nbytes, create_ms=1e3 * med("create"), upload_ms=1e3 * med("upload"),
           upload_GBps=nbytes / med("upload") / 1e9, first_solve_ms=1e3 * med("first"), problem_solve_ms=1e3 * med("later"),
           readback_cp2_ms=1e3 * med("readback"), evals=rows[0]["evals"],
           per_bucket_ms={k[3:]: round(1e3 * med(k), 3) for k in rows[0] if k.startswith("up_")},
           pcie_inclusive_subproblems_per_s=cfmm.problem.network_pool_count(net) * rows[0]["evals"] / (med("create") + med("upload") + med("first")))
print(json.dumps(out))
