#!/usr/bin/env python
"""B price vectors per pool read (cfmm_solve_batch): cost of a batched outer iteration against B.
For each B: `reps` cold batched solves of the same B utilities (arbitrage under perturbed market values) on one config;
one JSON line per B with wall ms per batch, device us per lock-step iteration, pool-subproblems/s (pools x sum of the
solves' evaluations / wall), and the single-solve path's figures for the same utilities as the reference line."""
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

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--sizes", default="1,2,4,8")
ap.add_argument("--tol", type=float, default=1e-6)
args = ap.parse_args()
net = synthetic.config(args.config, seed=0) if args.config != "C4shard" else synthetic.config("C4", scale=0.125, seed=0)
n = net["n_tokens"]
rng = np.random.default_rng(1)
utils = [cfmm.Arbitrage(net["c"] * np.exp(rng.normal(0, 0.01, n))) for _ in range(8)]
p = cfmm.Problem.from_network(net, utility=utils[0])
m = p.m
# reference: the single-solve path, one utility after the other
p.solve(tol=args.tol)
t0 = time.perf_counter(); ev1 = 0; dev1 = 0.0
for _ in range(args.reps):
    for u in utils:
        p.set_utility(u); p.solve(tol=args.tol)
        assert p.status == "optimal"
        ev1 += p.stats["evals"]; dev1 += p.stats["device_seconds"]
w1 = time.perf_counter() - t0
print(json.dumps(dict(config=args.config, B=0, path="single", solves=8 * args.reps, evals=ev1, wall_ms_per_solve=1e3 * w1 / (8 * args.reps),
                      device_us_per_eval=1e6 * dev1 / ev1, subproblems_per_s=m * ev1 / w1)), flush=True)
for B in [int(x) for x in args.sizes.split(",")]:
    B = min(B, p.ctx.batch_capacity())
    us = utils[:B]
    p.solve_many(us, tol=args.tol, batch=B)         # warm-up: clones, attributes
    t0 = time.perf_counter(); ev = 0; dev = 0.0; iters = 0
    for _ in range(args.reps):
        res = p.solve_many(us, tol=args.tol, batch=B)
        assert all(r["status"] == "optimal" for r in res), [r["status"] for r in res]
        ev += sum(r["stats"]["evals"] for r in res)
        iters += max(r["stats"]["evals"] for r in res)
        dev += res[0]["stats"]["device_seconds"]
    w = time.perf_counter() - t0
    print(json.dumps(dict(config=args.config, B=B, path="batch", solves=B * args.reps, evals=ev, lockstep_iterations=iters,
                          wall_ms_per_batch=1e3 * w / args.reps, device_us_per_iteration=1e6 * dev / iters,
                          device_us_per_solve_iteration=1e6 * dev / ev, subproblems_per_s=m * ev / w,
                          device_subproblems_per_s=m * ev / dev)), flush=True)
p.close()
