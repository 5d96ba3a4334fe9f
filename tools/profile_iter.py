#!/usr/bin/env python
"""Profiling target: `--solves` cold solves of a config and nothing else, so that a rocprofv3 per-kernel table over this
process shows the outer iteration as it runs in bench.py's timed region (iter_kernel: one launch per iteration).  The
last `run_ahead` launches of every solve return at once (the solve has ended): the MEDIAN, not the mean, is the launch.

    rocprofv3 --kernel-trace --stats -- python tools/profile_iter.py --config C3
    rocprofv3 --pmc FETCH_SIZE       -- python tools/profile_iter.py --config C3
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--zipf", type=float, default=None, help="token pairs Zipf(s) hub-weighted (SURVEY 8(d) stress variant)")
ap.add_argument("--solves", type=int, default=10)
args = ap.parse_args()

import cfmm  # noqa: E402
from cfmm import synthetic  # noqa: E402

net = synthetic.config(args.config, seed=0, zipf_s=args.zipf)
prob = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
ev = dev = 0
for _ in range(args.solves):
    prob.solve(tol=1e-6)
    ev += prob.stats["evals"]; dev += prob.stats["device_seconds"]
print(json.dumps(dict(config=args.config, pools=prob.m, solves=args.solves, evals_per_solve=ev / args.solves, status=prob.status,
                      device_us_per_iteration=1e6 * dev / ev, gap=prob.gap, infeas=prob.infeas)))
