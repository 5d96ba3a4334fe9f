"""fuzz_table.py's instances by seed through both outer iterations, with the second-order step trace:   python tools/table_seeds.py <seed> ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, cfmm
from helpers import problem_of, table_instance
if os.environ.get("TRACE", "1") != "0":
    os.environ["CFMM_NEWTON_TRACE"] = "1"
for seed in [int(a) for a in sys.argv[1:]]:
    inst, with_sum = table_instance(seed)
    p = problem_of(inst)
    v = p.solve(tol=1e-8)
    print("seed", seed, "n", inst["n_tokens"], "pools", len(inst["kinds"]), "sum", with_sum, inst["utility"]["type"], "default:", p.status, v, p.gap, p.infeas, p.stats.get("method"), flush=True)
    v2 = p.solve(tol=1e-7, method="newton")
    print("   newton:", p.status, v2, p.gap, p.infeas, p.stats.get("newton_steps"), flush=True)
    p.close()
