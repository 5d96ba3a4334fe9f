#!/usr/bin/env python
"""K-asset table buckets: evaluation time of the table's own launch (table_eval_kernel) for m stableswap pools of k assets,
cold (no warm starts: CFMM_TABLE_WARM=0 semantics via fresh prices) and warm (the previous evaluation's roots), + parity of one
evaluation against the NumPy restatement.   python tools/table_timing.py [m] [k]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    sys.path.insert(0, p)
import numpy as np
import cfmm
from cfmm import synthetic, _lib
from oracle import pools_np
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
out = {}
for kind in ("stable", "sum"):
    net = synthetic.make_network(1000, m_cp2=1000, seed=3, **{f"m_gk_{kind}": m}, gk_sizes=(k, k))
    n = net["n_tokens"]
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    rng = np.random.default_rng(1)
    nu = net["c"] * np.exp(rng.normal(0, 0.01, n))
    f, psi = p.eval_dual(nu)
    b = net["gk"][(kind, k)]
    if kind == "stable" and m <= 20000:
        y, arb = pools_np.arb_stable_n(b["R"], b["param"], b["fee"], nu[b["idx"]])
        ref = np.zeros(n); np.add.at(ref, b["idx"].ravel(), y.ravel())
        q = cfmm.Problem.from_network({kk: v for kk, v in net.items() if kk != "gk"}, utility=cfmm.Arbitrage(net["c"]))
        f0, psi0 = q.eval_dual(nu); q.close()
        out[f"{kind}_parity_rel"] = float(np.abs(psi - psi0 - ref).max() / np.abs(ref).max())
    res = {}
    for tag, warm in (("cold", "0"), ("warm", "1")):
        os.environ["CFMM_TABLE_WARM"] = warm
        p.ctx.set_nu(nu)
        res[tag + "_us"] = 1e6 * p.ctx.time_eval_kernel(_lib.TIME_TABLE, 50)
    # warm starts under moving prices: a new price vector 0.1 % away per launch
    p.ctx.set_nu(nu * np.exp(rng.normal(0, 1e-3, n)))
    res["warm_after_0.1pct_move_us"] = 1e6 * p.ctx.time_eval_kernel(_lib.TIME_TABLE, 1)
    res["pools"] = m; res["assets"] = k; res["trading_fraction"] = float((np.abs(psi) > 0).mean())
    out[kind] = res
    p.close()
print(json.dumps(out))
