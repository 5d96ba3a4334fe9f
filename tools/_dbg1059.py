import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, cfmm
from helpers import problem_of, table_instance, normalise_with_params
from oracle import dual_np
np.set_printoptions(precision=6, linewidth=200)
for seed in [int(a) for a in sys.argv[1:]]:
    inst, with_sum = table_instance(seed)
    p = problem_of(inst)
    v = p.solve(tol=1e-8)
    d = dual_np.solve_dual(normalise_with_params(inst))
    print("seed", seed, "status", p.status, "value", v, "dual bound", d["value"], "gap", p.gap, "infeas", p.infeas, "stats", {k: p.stats.get(k) for k in ("evals", "method", "rounds", "newton_steps")})
    print(" nu", p.nu, "\n ref nu", d["nu"])
    print(" psi", p.psi)
    print(" theta records:", [(k, {a: b for a, b in rec.items() if a in ("sgn", "ia", "ib", "fee", "Ra", "Rb", "leg_lo", "pidx")}, th) for k, (rec, th) in p._theta.items()])
    for i, (li, R, g, kind, prm, dd, ll) in enumerate(zip(inst["local_indices"], inst["reserves"], inst["fees"], inst["kinds"], inst["params"], p.deltas, p.lambdas)):
        x = np.asarray(R) + g * dd - ll
        flag = "  <-- NEGATIVE" if (np.any(dd < 0) or np.any(ll < 0) or np.any(x < -1e-9 * np.max(R))) else ""
        if kind == "sum" or flag:
            print("  pool", i, kind, "tokens", li, "R", np.asarray(R), "fee", g, "delta", dd, "lambda", ll, "post", x, "prices", p.nu[li], flag)
    v2 = p.solve(tol=1e-8, method="newton")
    print(" newton:", p.status, v2, p.gap, p.infeas)
    p.close()
