#!/usr/bin/env python
"""TEST INFRASTRUCTURE -- generates tests/golden/c5_liquidation.json: BASELINE config 5 (5e5 stableswap + 5e4 constant-product
pools, 1000 tokens; the liquidation of a 10-token basket, /root/reference/liquidation.py:57,77-80) solved at FULL SIZE by the
independent NumPy second-order solver of oracle/barrier_newton.py, certificates from the C oracle's exact dual evaluation.
No SciPy primal reaches this size and the C oracle's first-order iteration is 0.4% from the optimum after 2000 evaluations,
so this is the only CPU solve the HIP library's config-5 optimum can be compared with (tests/test_gpu_newton.py).

    python oracle/make_c5_fixture.py            # ~30 minutes on 8 cores (~65 smoothed evaluations of 1.1e6 pool directions)

The instance is rebuilt from seeds by the test (cfmm.synthetic.config("C5"), basket seed 1): the fixture holds only the
solver's results and its final prices (at which the test also compares the exact dual evaluations).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")]
from cfmm import synthetic                      # noqa: E402  (the instance generator only; nothing is solved by the product here)
from oracle import barrier_newton, c_oracle     # noqa: E402

GE, EQ, FREE = 0, 1, 2


def basket(net, seed=1, k=10):
    """the basket of tests/test_gpu_newton.py::_basket and bench.py's config 5"""
    n = net["n_tokens"]
    rng = np.random.default_rng(seed)
    h = np.zeros(n); idx = rng.choice(n, min(k, n - 1), replace=False)
    h[idx] = np.exp(rng.normal(2, 0.5, len(idx))) / net["prices"][idx] * 10
    tgt = int(rng.integers(0, n)); h[tgt] = 0
    return h, tgt


def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    net = synthetic.config("C5", scale=scale) if scale != 1.0 else synthetic.config("C5")
    n = net["n_tokens"]
    h, t = basket(net)
    c = np.zeros(n); c[t] = 1.0                                   # liquidation.py:57   max psi[t]
    ct = np.full(n, EQ, dtype=np.int32); ct[t] = FREE             # liquidation.py:77-80   psi + h == 0 off the target
    O = c_oracle.Oracle(n, threads=os.cpu_count()); O.add_network(net); O.set_utility(c, h, ct)
    t0 = time.time()
    r = barrier_newton.solve(net, c, h, ct, tol=2e-7, exact_eval=lambda nu: O.eval(nu),
                             log=lambda m: print("[%.0fs] %s" % (time.time() - t0, m), file=sys.stderr, flush=True))
    out = dict(instance="cfmm.synthetic.config('C5'%s), basket(seed=1, k=10), Liquidate" % ("" if scale == 1.0 else ", scale=%g" % scale),
               solver="oracle/barrier_newton.py (NumPy barrier path-following, LAPACK Cholesky); certificates: oracle/cfmm_oracle.c exact dual",
               pools=int(sum(len(net[k]["Ra"]) for k in ("cp2", "w2", "curve2", "pow2", "sum2") if k in net)), tokens=n, target=t,
               dual_value=r["dual_value"], primal_value=r["primal_value"], gap=r["gap"], infeas=r["infeas"], steps=r["steps"],
               smoothed_evaluations=r["evals"], barrier_mu=r["mu"], seconds=round(time.time() - t0, 1),
               nu=[float(x) for x in r["nu"]])
    path = os.path.join(ROOT, "tests", "golden", "c5_liquidation.json" if scale == 1.0 else "c5_liquidation_scale%g.json" % scale)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
