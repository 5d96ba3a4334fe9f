"""fuzz_mid.py's instances by seed through the outer iterations, with the second-order step trace:   python tools/mid_seeds.py <seed> ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")): sys.path.insert(0, p)
import numpy as np, cfmm
from cfmm import synthetic


def instance(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(50, 400))
    kw = dict(m_cp2=int(rng.integers(500, 20000)), m_w2=int(rng.integers(0, 3000)), m_gn=int(rng.integers(0, 2000)),
              m_curve2=int(rng.integers(0, 3000)) * int(rng.random() < 0.5), m_gk_stable=int(rng.integers(0, 1500)) * int(rng.random() < 0.5),
              m_gk_sum=int(rng.integers(0, 300)) * int(rng.random() < 0.4), m_pow2=int(rng.integers(0, 1000)) * int(rng.random() < 0.3))
    net = synthetic.make_network(n, seed=seed, **kw)
    ut = seed % 3
    h = np.zeros(n); basket = rng.choice(n, 8, replace=False); t = int(basket[0]); h[basket[1:]] = 20.0 / net["prices"][basket[1:]]
    util = cfmm.Arbitrage(net["c"]) if ut == 0 else (cfmm.Swap(h, t) if ut == 1 else cfmm.Liquidate(h, t))
    return net, util, kw


if __name__ == "__main__":
    if os.environ.get("TRACE", "1") != "0":
        os.environ["CFMM_NEWTON_TRACE"] = "1"
    for seed in [int(a) for a in sys.argv[1:]]:
        net, util, kw = instance(seed)
        p = cfmm.Problem.from_network(net, utility=util)
        for m in ("lbfgs", "newton"):
            v = p.solve(tol=1e-6, max_evals=6000, method=m)
            print("seed", seed, m, p.status, v, p.gap, p.infeas, {k: p.stats.get(k) for k in ("evals", "method", "newton_steps", "rounds")}, flush=True)
        p.close()
