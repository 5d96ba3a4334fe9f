#!/usr/bin/env python
"""Fuzz campaign over small random instances in the reference's vocabulary (tests/helpers.py: random_instance): every instance through
(a) Problem.solve (tiny path + host kink loop), (b) the same through the swept call with a dozen scaled utilities, (c) the second-order
method, and against the SciPy primal.  Prints one line per failure and a summary.   python tools/fuzz_small.py [first_seed] [count]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import cfmm
from helpers import random_instance, problem_of, normalise_with_params, utility_of
from oracle.primal_scipy import solve_primal
from oracle import dual_np

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fails, stats = [], dict(n=0, infeasible=0, newton_checked=0, swept=0, slsqp_fail=0, independent_value=0, referee_loose=0)
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    util = ["arbitrage", "swap", "liquidate"][seed % 3]
    kw = dict(n_tokens=int(rng.integers(3, 9)), n_pools=int(rng.integers(4, 24)), with_sum=bool(seed % 2), with_curve=bool((seed // 2) % 2),
              with_power=bool((seed // 4) % 3 == 0), utility=util)
    inst = random_instance(seed, **kw)
    tag = f"seed {seed} {kw}"
    try:
        p = problem_of(inst)
        v = p.solve(tol=1e-9)
        stats["n"] += 1
        r = solve_primal(normalise_with_params(inst))
        if p.status == "infeasible":
            stats["infeasible"] += 1
            if r["success"]:
                fails.append(f"{tag}: infeasible here, SLSQP value {r['value']}")
            p.close(); continue
        if not (p.status == "optimal" and p.gap <= 1e-8 and p.infeas <= 1e-8):
            fails.append(f"{tag}: status {p.status} gap {p.gap:.2e} infeas {p.infeas:.2e} evals {p.stats['evals']}")
            p.close(); continue
        if not r["success"]:
            stats["slsqp_fail"] += 1
        # the referee that answers (round 6): the dual of the decomposed program minimised by SciPy over the NumPy pool restatements
        # (oracle/dual_np.py).  Its value is an upper bound on the optimum whatever happens: a "certified" value ABOVE it is a false
        # certificate; a value within 2e-6 below it is an independently confirmed optimum.
        d = dual_np.solve_dual(normalise_with_params(inst))
        tolv = 2e-6 * max(1.0, abs(v))
        if v > d["value"] + tolv:
            fails.append(f"{tag}: FALSELY CERTIFIED: value {v} above the independent dual bound {d['value']}")
        elif d["value"] - v <= tolv or (r["success"] and abs(r["value"] - v) <= tolv):
            stats["independent_value"] += 1
        else:
            stats["referee_loose"] += 1                          # (the referee stopped short: nothing learnt about this instance)
        if r["success"] and r["value"] > v + 2e-6 * max(1, abs(v)):
            fails.append(f"{tag}: SLSQP found a BETTER primal point {r['value']} > {v}")
        elif r["success"] and r["value"] < v - 2e-6 * max(1, abs(v)):
            stats["slsqp_worse"] = stats.get("slsqp_worse", 0) + 1        # (a certified optimum above SLSQP's point: SLSQP stopped early)
        # tenders add up to psi, are complementary and non-negative
        tot = np.zeros(inst["n_tokens"])
        for li, dd, ll in zip(inst["local_indices"], p.deltas, p.lambdas):
            np.add.at(tot, li, ll - dd)
            if np.any(dd < 0) or np.any(ll < 0):
                fails.append(f"{tag}: negative tender")
        if np.abs(tot - p.psi).max() > 1e-7 * max(1.0, np.abs(p.psi).max()):
            fails.append(f"{tag}: tenders do not add up to psi ({np.abs(tot - p.psi).max():.2e})")
        # second order on the same instance
        try:
            v2 = p.solve(tol=1e-8, method="newton")
            stats["newton_checked"] += 1
            if not (p.status == "optimal" and abs(v2 - v) <= 1e-6 * max(1, abs(v))):
                fails.append(f"{tag}: newton {p.status} {v2} vs {v} (gap {p.gap:.1e} infeas {p.infeas:.1e}, {p.stats.get('newton_steps')} steps)")
        except cfmm.CfmmError as e:
            if "cannot take" not in str(e) and "unsupported" not in str(e).lower():
                fails.append(f"{tag}: newton raised {e}")
        # the swept call over scaled utilities (linear-box utilities, no stableswap / power-sum pools)
        if not (kw["with_curve"] or kw["with_power"]):
            u0 = utility_of(inst)
            utils = []
            for k in range(8):
                if util == "arbitrage":
                    utils.append(cfmm.Arbitrage(u0.c * np.exp(rng.normal(0, 0.03 * (1 + k), inst["n_tokens"]))))
                else:
                    h = u0.h * (0.2 + 0.6 * k)
                    utils.append(cfmm.Swap(h, inst["utility"]["t"]) if util == "swap" else cfmm.Liquidate(h, inst["utility"]["t"]))
            res = p.solve_many(utils, tol=1e-9)
            stats["swept"] += 1
            q = problem_of(inst)
            for k, (u, rr) in enumerate(zip(utils, res)):
                q.set_utility(u)
                vq = q.solve(tol=1e-9)
                if rr["status"] != q.status:
                    fails.append(f"{tag}: sweep point {k} status {rr['status']} vs {q.status}")
                elif q.status == "optimal" and abs(rr["value"] - vq) > 1e-7 * max(1, abs(vq)):
                    fails.append(f"{tag}: sweep point {k} value {rr['value']} vs {vq}")
            q.close()
        p.close()
    except Exception as e:                                 # noqa: BLE001 -- a fuzz run reports and goes on
        fails.append(f"{tag}: EXCEPTION {type(e).__name__}: {e}")
print(json.dumps(dict(stats, seconds=round(time.time() - t0, 1), failures=len(fails))))
for f in fails:
    print("FAIL", f)
