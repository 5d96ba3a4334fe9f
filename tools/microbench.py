#!/usr/bin/env python
"""Kernel-tuning probe (not part of the product, not part of bench.py's contract): one JSON line with
the fused evaluation kernel's launch time (all buckets / per bucket) and the per-iteration cost of
full solves, for whatever library build CFMM_LIB points at and whatever CFMM_* knobs are set.

    CFMM_LIB=.../variants/libcfmm_hip_x.so CFMM_SLICES=32 python tools/microbench.py --config C3 --tag x
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--tag", default="")
ap.add_argument("--solves", type=int, default=5)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--buckets", action="store_true")
ap.add_argument("--many", type=int, default=0, help="also time N solves through solve_many at concurrency 1, 2, 3")
args = ap.parse_args()

import numpy as np  # noqa: E402
import cfmm  # noqa: E402
from cfmm import synthetic, _lib  # noqa: E402
import bench  # noqa: E402

net = synthetic.config(args.config, seed=0, scale=args.scale)
prob = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
prob.solve(tol=1e-6)
out = dict(tag=args.tag, lib=os.path.basename(os.environ.get("CFMM_LIB", "default")),
           env={k: v for k, v in os.environ.items() if k.startswith("CFMM_") and k != "CFMM_LIB"},
           config=args.config, pools=prob.m, status=prob.status, evals=prob.stats["evals"], value=prob.value,
           gap=prob.gap, infeas=prob.infeas)
ev = dev = wall = 0
t0 = time.perf_counter()
for _ in range(args.solves):
    prob.solve(tol=1e-6)
    ev += prob.stats["evals"]; dev += prob.stats["device_seconds"]; wall += prob.stats["wall_seconds"]
out["solve_wall_ms"] = 1e3 * (time.perf_counter() - t0) / args.solves
out["dev_us_per_eval"] = 1e6 * dev / ev
out["wall_us_per_eval"] = 1e6 * wall / ev
out["eval_all_us"] = 1e6 * prob.ctx.time_eval_kernel(_lib.TIME_ALL, args.reps)
if args.many:
    rng = np.random.default_rng(5)
    utils = [cfmm.Arbitrage(net["c"] * np.exp(rng.normal(0, 0.005, net["n_tokens"]))) for _ in range(args.many)]
    out["solve_many"] = {}
    for conc in (1, 2, 3, 4):
        prob.solve_many(utils[:4], concurrency=conc, tol=1e-6)
        t0 = time.perf_counter()
        res = prob.solve_many(utils, concurrency=conc, tol=1e-6)
        dt = time.perf_counter() - t0
        ev = sum(r["stats"]["evals"] for r in res)
        out["solve_many"][conc] = dict(ms_per_solve=round(1e3 * dt / len(utils), 3), pool_subproblems_per_s=ev * prob.m / dt,
                                       all_optimal=all(r["status"] == "optimal" for r in res))
if args.buckets:
    out["buckets"] = {r["kernel"]: round(r["us"], 2) for r in bench.kernel_table(prob, args.reps)[1:]}
prob.ctx.debug_timers()            # clears the logs
prob.eval_dual(prob.nu)            # ONE launch: its in-kernel span (first block start .. last block end)
_, _, tb = prob.ctx.debug_timers()
prob.ctx.time_eval_kernel(_lib.TIME_ALL, 20)
ts, tl, _ = prob.ctx.debug_timers()
if ts.any():
    names = ["gn8", "gn7", "gn6", "gn5", "gn4", "gn3", "curve2", "w2", "cp2", "sum2"]
    bk = (tl >> 48) - 1
    cyc = tl & ((1 << 48) - 1)
    out["tile_us(avg,max,count)"] = {nm: (round(float(cyc[bk == q].mean()) / 2400.0, 2), round(float(cyc[bk == q].max()) / 2400.0, 2),
                                          int((bk == q).sum())) for q, nm in enumerate(names) if (bk == q).any()}
    out["between_tiles_us(mean,max)"] = [round(float(cyc[:, 7][bk[:, 7] == 14].mean()) / 2400.0, 2), round(float(cyc[:, 7].max()) / 2400.0, 2)]
    cyc = cyc[:, :7]; bk = bk[:, :7]
    per_wave = cyc.sum(axis=1) / 2400.0
    out["wave_busy_us(min,mean,max)"] = [round(float(x), 2) for x in (per_wave[per_wave > 0].min(), per_wave[per_wave > 0].mean(), per_wave.max())]
    prob.solve(tol=1e-6, max_evals=12, method="lbfgs")
    tu, _, tbi = prob.ctx.debug_timers()
    d = lambda t, i, j: (int(t[j, 0] - t[i, 0]), round((t[j, 1] - t[i, 1]) * 0.01, 2))   # (cycles, us)
    tb = tb[tb[:, 1] > 0]
    t0 = tb[:, 0].min()
    st_, en_ = (tb[:, 0] - t0) * 0.01, (tb[:, 1] - t0) * 0.01
    out["eval_blocks"] = dict(n=len(tb), span_us=round(float(en_.max()), 2), start_pct=[round(float(x), 2) for x in np.percentile(st_, [0, 50, 90, 100])],
                              end_pct=[round(float(x), 2) for x in np.percentile(en_, [0, 10, 50, 90, 100])],
                              dur_pct=[round(float(x), 2) for x in np.percentile(en_ - st_, [0, 50, 90, 100])])
    out["eval_phases(cyc,us)"] = dict(prologue=d(ts, 0, 1), tiles=d(ts, 1, 2), reduce=d(ts, 2, 3), flush=d(ts, 3, 4))
    out["upd_phases(cyc,us)"] = dict(st=d(tu, 8, 9), loads=d(tu, 9, 10), A=d(tu, 10, 11), B=d(tu, 11, 12), C=d(tu, 12, 13),
                                     D=d(tu, 13, 14), E=d(tu, 14, 15), total=d(tu, 8, 15))
    if tu[16].any():
        out["iter_phases(cyc,us)"] = dict(st=d(tu, 16, 17), loads_grad=d(tu, 17, 18), gram_reduce=d(tu, 18, 19), lds_exchange=d(tu, 19, 20),
                                          recursion=d(tu, 20, 21), direction_F=d(tu, 21, 22), trial=d(tu, 22, 23), tiles=d(tu, 23, 2),
                                          reduce=d(tu, 2, 3), flush=d(tu, 3, 4), total=d(tu, 16, 4))
    if tu[16].any() and tbi[:256, 0].all():
        # the last full launch of that solve, per workgroup (100 MHz wall clock): start | update done | end
        st0 = tbi[:256, 0].min()
        b_st, b_up, b_en = (tbi[:256, 0] - st0) * 0.01, (tbi[256:512, 0] - st0) * 0.01, (tbi[:256, 1] - st0) * 0.01
        pc = lambda x: [round(float(v), 2) for v in np.percentile(x, [0, 10, 50, 90, 100])]
        out["iter_blocks"] = dict(start=pc(b_st), update_done=pc(b_up), end=pc(b_en), update_us=pc(b_up - b_st), tiles_us=pc(b_en - b_up),
                                  wg0=[round(float(b_st[0]), 2), round(float(b_up[0]), 2), round(float(b_en[0]), 2)],
                                  slowest_end_wg=int(np.argmax(b_en)), slowest_update_wg=int(np.argmax(b_up)))
    if tu[19].any() and not tu[16].any():
        out["upd_A_detail(cyc,us)"] = dict(grad=d(tu, 10, 19), partials=d(tu, 19, 20), reduce_scatter=d(tu, 20, 21), lds_exchange=d(tu, 21, 22), tail=d(tu, 22, 11))
print(json.dumps(out), flush=True)
