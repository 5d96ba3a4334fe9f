"""explicit second-order solves of fuzz_small.py's instances by seed, with the step trace:   python tools/newton_seeds.py <seed> ...
(tools/stress_shared_gpu.sh runs six of these side by side on one device; tests/test_gpu_newton.py keeps a smaller copy of that run)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, cfmm
from helpers import random_instance, problem_of
os.environ["CFMM_NEWTON_TRACE"] = "1"
for seed in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(seed)
    util = ["arbitrage", "swap", "liquidate"][seed % 3]
    kw = dict(n_tokens=int(rng.integers(3, 9)), n_pools=int(rng.integers(4, 24)), with_sum=bool(seed % 2), with_curve=bool((seed // 2) % 2),
              with_power=bool((seed // 4) % 3 == 0), utility=util)
    inst = random_instance(seed, **kw)
    p = problem_of(inst)
    v = p.solve(tol=1e-9)
    print("seed", seed, kw, "default:", p.status, v, p.gap, p.infeas, flush=True)
    try:
        v2 = p.solve(tol=1e-8, method="newton")
        print("   newton:", p.status, v2, p.gap, p.infeas, p.stats.get("newton_steps"), p.stats.get("numeric_error"), flush=True)
    except Exception as e:
        print("   newton raised", e)
    print("   start prices", cfmm.start_prices(p.net, p.utility), "nu", p.nu)
    p.close()
