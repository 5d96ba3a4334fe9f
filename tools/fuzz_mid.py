#!/usr/bin/env python
"""Fuzz campaign over MID-SIZE random networks (50-400 tokens, 1e3-3e4 pools of every kind the library holds, in random proportions,
under the three utilities): the default path, method="lbfgs" and method="newton" against each other and against their own certificates.
    python tools/fuzz_mid.py [first_seed] [count]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    sys.path.insert(0, p)
import numpy as np
import cfmm
from cfmm import synthetic

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fails, stats = [], dict(n=0, newton=0)
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(50, 400))
    kw = dict(m_cp2=int(rng.integers(500, 20000)), m_w2=int(rng.integers(0, 3000)), m_gn=int(rng.integers(0, 2000)),
              m_curve2=int(rng.integers(0, 3000)) * int(rng.random() < 0.5), m_gk_stable=int(rng.integers(0, 1500)) * int(rng.random() < 0.5),
              m_gk_sum=int(rng.integers(0, 300)) * int(rng.random() < 0.4), m_pow2=int(rng.integers(0, 1000)) * int(rng.random() < 0.3))
    net = synthetic.make_network(n, seed=seed, **kw)
    ut = seed % 3
    if ut == 0:
        util = cfmm.Arbitrage(net["c"])
    else:
        h = np.zeros(n); basket = rng.choice(n, 8, replace=False); t = int(basket[0]); h[basket[1:]] = 20.0 / net["prices"][basket[1:]]
        util = cfmm.Swap(h, t) if ut == 1 else cfmm.Liquidate(h, t)
    tag = f"seed {seed} n {n} {kw} util {('arb', 'swap', 'liq')[ut]}"
    try:
        p = cfmm.Problem.from_network(net, utility=util)
        v = p.solve(tol=1e-6, max_evals=6000)
        stats["n"] += 1
        if not (p.status == "optimal" and p.gap <= 1e-6 and p.infeas <= 1e-6):
            fails.append(f"{tag}: auto {p.status} gap {p.gap:.2e} infeas {p.infeas:.2e} evals {p.stats['evals']} method {p.stats.get('method')} steps {p.stats.get('newton_steps')}")
            p.close(); continue
        try:
            v2 = p.solve(tol=1e-6, method="newton")
            stats["newton"] += 1
            if not (p.status == "optimal" and abs(v2 - v) <= 4e-6 * max(1.0, abs(v))):
                fails.append(f"{tag}: newton {p.status} {v2} vs {v} (gap {p.gap:.1e} infeas {p.infeas:.1e}, {p.stats.get('newton_steps')} steps)")
        except cfmm.CfmmError as e:
            if "cannot take" not in str(e):
                fails.append(f"{tag}: newton raised {e}")
        p.close()
    except Exception as e:                                 # noqa: BLE001
        fails.append(f"{tag}: EXCEPTION {type(e).__name__}: {e}")
print(json.dumps(dict(stats, seconds=round(time.time() - t0, 1), failures=len(fails))))
for f in fails:
    print("FAIL", f)
