#!/usr/bin/env python
"""Profiling target: uploads a synthetic config and launches the fused evaluation kernel `--reps` times
back to back (cfmm_time_eval_kernel, 3 warm-up launches first) -- nothing else runs an eval_kernel<false>,
so a rocprofv3 per-kernel average over this process IS the steady-state launch.  Prints one JSON line.

    rocprofv3 --kernel-trace --stats -- python tools/profile_eval.py --config C3
    rocprofv3 --pmc FETCH_SIZE       -- python tools/profile_eval.py --config C3
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
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--only", type=int, default=None, help="one bucket alone: 0 cp2, 1 w2, 2 sum2, 3 curve2, -k the k-asset bucket")
args = ap.parse_args()

import numpy as np  # noqa: E402
import cfmm  # noqa: E402
from cfmm import synthetic, _lib  # noqa: E402
import bench  # noqa: E402

net = synthetic.config(args.config, seed=0, scale=args.scale, zipf_s=args.zipf)
prob = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
ctx = prob._ensure_ctx()
# prices a few percent off the market values: ~88 % of the pools trade, as at the first iterations of a solve
ctx.set_nu(net["c"] * np.exp(np.random.default_rng(0).normal(0, 0.01, net["n_tokens"])))
sec = ctx.time_eval_kernel(_lib.TIME_ALL if args.only is None else args.only, args.reps)
nbytes = sum(len(net[k]["Ra"]) * bench.BYTES_PER_POOL[k] for k in ("cp2", "w2", "sum2", "curve2") if k in net)
nbytes += sum(b["R"].shape[1] * (20 + 20 * k) for k, b in net.get("gn", {}).items())
print(json.dumps(dict(config=args.config, pools=prob.m, tokens=net["n_tokens"], reps=args.reps, launch_us=sec * 1e6,
                      algorithmic_bytes=nbytes, GBps=nbytes / sec / 1e9, pools_per_s=prob.m / sec)))
