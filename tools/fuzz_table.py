#!/usr/bin/env python
"""Fuzz campaign for the K-asset trading-function table: small instances with n-asset stableswap pools (k = 2..5 tokens of a peg group,
random amplification, random fee) and n-asset constant-sum pools among constant-product pools, under the three utilities of the
reference: first-order path, second-order path (stableswap only) and the SciPy primal with the same phi against each other.
    python tools/fuzz_table.py [first_seed] [count]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import cfmm
from helpers import problem_of, normalise_with_params
from oracle.primal_scipy import solve_primal
from oracle import dual_np


from helpers import table_instance as instance


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fails, stats = [], dict(n=0, newton=0, slsqp_ok=0, infeasible=0, independent_value=0, referee_loose=0)
t0 = time.time()
for seed in range(first, first + count):
    inst, with_sum = instance(seed)
    tag = f"seed {seed} n {inst['n_tokens']} pools {len(inst['kinds'])} sum {with_sum} {inst['utility']['type']}"
    try:
        p = problem_of(inst)
        v = p.solve(tol=1e-8)
        stats["n"] += 1
        r = solve_primal(normalise_with_params(inst))
        stats["slsqp_ok"] += bool(r["success"])
        if p.status == "infeasible":
            stats["infeasible"] += 1
            if r["success"]:
                fails.append(f"{tag}: infeasible here, SLSQP {r['value']}")
            p.close(); continue
        if not (p.status == "optimal" and p.gap <= 1e-8 and p.infeas <= 1e-8):
            fails.append(f"{tag}: status {p.status} gap {p.gap:.2e} infeas {p.infeas:.2e} evals {p.stats['evals']} method {p.stats.get('method')}")
            p.close(); continue
        if r["success"] and r["value"] > v + 2e-6 * max(1, abs(v)):
            fails.append(f"{tag}: SLSQP found a BETTER primal point {r['value']} > {v}")
        # the referee that answers (round 6, oracle/dual_np.py): an independent upper bound on the optimum, tight where it converges
        d = dual_np.solve_dual(normalise_with_params(inst))
        tolv = 2e-6 * max(1.0, abs(v))
        if v > d["value"] + tolv:
            fails.append(f"{tag}: FALSELY CERTIFIED: value {v} above the independent dual bound {d['value']}")
        elif d["value"] - v <= tolv or (r["success"] and abs(r["value"] - v) <= tolv):
            stats["independent_value"] += 1
        else:
            stats["referee_loose"] += 1
        tot = np.zeros(inst["n_tokens"])
        for li, R, g, kind, prm, dd, ll in zip(inst["local_indices"], inst["reserves"], inst["fees"], inst["kinds"], inst["params"], p.deltas, p.lambdas):
            np.add.at(tot, li, ll - dd)
            x = np.asarray(R) + g * dd - ll
            if kind == "curve":
                phi = lambda z: z.sum() - prm / np.prod(z)
                if phi(x) < phi(np.asarray(R)) - 1e-7 * np.sum(R):
                    fails.append(f"{tag}: a stableswap pool ends below its level set")
            if np.any(dd < 0) or np.any(ll < 0) or np.any(x < -1e-9 * np.max(R)):
                fails.append(f"{tag}: negative tender / reserve")
        if np.abs(tot - p.psi).max() > 1e-7 * max(1.0, np.abs(p.psi).max()):
            fails.append(f"{tag}: tenders do not add up to psi ({np.abs(tot - p.psi).max():.2e})")
        if not with_sum:
            v2 = p.solve(tol=1e-7, method="newton")
            stats["newton"] += 1
            if not (p.status == "optimal" and abs(v2 - v) <= 1e-6 * max(1, abs(v))):
                fails.append(f"{tag}: newton {p.status} {v2} vs {v} (gap {p.gap:.1e} infeas {p.infeas:.1e}, {p.stats.get('newton_steps')} steps)")
        p.close()
    except Exception as e:                                 # noqa: BLE001
        fails.append(f"{tag}: EXCEPTION {type(e).__name__}: {e}")
print(json.dumps(dict(stats, seconds=round(time.time() - t0, 1), failures=len(fails))))
for f in fails:
    print("FAIL", f)
