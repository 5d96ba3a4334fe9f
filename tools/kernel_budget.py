#!/usr/bin/env python
"""Kernel-time regression guard (profiling as an aux subsystem).

    python tools/kernel_budget.py --write     # on an MI355X: measure, write profiles/budget.json (measured + 10 %)
    python tools/kernel_budget.py             # measure and compare with the committed budget (what tests/test_gpu_perf.py does)

Every entry is the time of ONE dominant kernel (or launch group) of a BASELINE configuration, taken with the library's own
timing hooks (HIP events around back-to-back launches on its stream) as the MINIMUM over `--rounds` rounds of a mean over
`reps` launches: a shared box's hiccups only ever make a round slower, so the minimum is the kernel; a real regression
(round 3's misaligned exchange strips: eval_batch_kernel 68 -> 139 us, which no test could see) moves it.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
BUDGET = os.path.join(ROOT, "profiles", "budget.json")
MARGIN = 1.12
# launch chains (the factorisation: 17 dependent launches of ~13 us of chain each) carry the BOX's dispatch latency 17 times: five
# boxes of one round measured 313 ... 357 us for the same binary, where a single-launch kernel moves by 2-4 %
MARGIN_CHAIN = 1.25
def margin(key):
    return MARGIN_CHAIN if ("factor" in key or "sweep" in key or "ulog" in key) else MARGIN        # (launch chains and host-paced loops)


def measure(rounds=5, reps=20, only=None):
    import numpy as np
    import cfmm
    from cfmm import synthetic, _lib
    out = {}

    def best(f):
        return min(f() for _ in range(rounds))

    def first_order(cfg, batch=False):
        net = synthetic.config(cfg, seed=0)
        prob = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
        prob.solve(tol=1e-6)
        assert prob.status == "optimal", (cfg, prob.status)

        def iteration():                                   # iter_kernel: device time of a cold solve / its launches
            prob.solve(tol=1e-6)
            return 1e6 * prob.stats["device_seconds"] / prob.stats["evals"]
        out[f"{cfg}.iter_kernel_us_per_iteration"] = best(iteration)
        prob.ctx.set_nu(net["c"] * np.exp(np.random.default_rng(0).normal(0, 0.01, net["n_tokens"])))
        out[f"{cfg}.eval_kernel_us"] = best(lambda: 1e6 * prob.ctx.time_eval_kernel(_lib.TIME_ALL, reps))
        if batch:
            B = prob.ctx.batch_capacity()
            rng = np.random.default_rng(1)
            us = [cfmm.Arbitrage(net["c"] * np.exp(rng.normal(0, 0.01, net["n_tokens"]))) for _ in range(B)]
            prob.solve_many(us, tol=1e-6, batch=B)

            def lockstep():                                # eval_batch_kernel + B update workgroups per lock-step iteration
                res = prob.solve_many(us, tol=1e-6, batch=B)
                return 1e6 * res[0]["stats"]["device_seconds"] / max(r["stats"]["evals"] for r in res)
            out[f"{cfg}.batch{B}_us_per_lockstep_iteration"] = best(lockstep)
        prob.close()

    def second_order():
        net = synthetic.config("C5", seed=0)
        rng = np.random.default_rng(1)
        n = net["n_tokens"]
        h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
        t_out = int(rng.integers(0, n)); h[t_out] = 0
        prob = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t_out))
        prob.solve(tol=1e-6)
        assert prob.status == "optimal", prob.status
        mu = max(prob.stats.get("barrier_mu", 0.0), 1e-12)
        rows = [prob.ctx.time_newton_kernels(mu, 5) for _ in range(rounds)]
        for k in ("smooth_hess", "smooth"):
            out[f"C5.{k}_us"] = 1e6 * min(r[k] for r in rows)
        # factorisation (with the inverse factor riding along) + the solve it leaves: the linear algebra of one Newton step
        out["C5.factor_plus_backsolve_us"] = 1e6 * min(r["factor"] + r["backsolve"] for r in rows)
        prob.ctx.set_nu(net["prices"])
        out["C5.eval_kernel_us"] = best(lambda: 1e6 * prob.ctx.time_eval_kernel(_lib.TIME_ALL, reps))
        prob.close()

    def table():
        # the K-asset table's own launch (csrc/phik.hpp: table_eval_kernel): 1e5 four-asset stableswap pools, the root searches warm-started
        # from the previous evaluation (what every evaluation of a solve but its first sees) and cold (CFMM_TABLE_WARM=0 in a child process)
        import subprocess
        net = synthetic.make_network(1000, m_cp2=1000, seed=3, m_gk_stable=100000, gk_sizes=(4, 4))
        prob = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
        prob._send_utility()
        prob.ctx.set_nu(net["c"] * np.exp(np.random.default_rng(1).normal(0, 0.01, net["n_tokens"])))
        out["table.stable4_1e5_warm_us"] = best(lambda: 1e6 * prob.ctx.time_eval_kernel(_lib.TIME_TABLE, reps))
        prob.close()
        code = ("import sys, json; sys.path[:0] = %r; import numpy as np, cfmm; from cfmm import synthetic, _lib\n"
                "net = synthetic.make_network(1000, m_cp2=1000, seed=3, m_gk_stable=100000, gk_sizes=(4, 4))\n"
                "p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net['c'])); p._send_utility()\n"
                "p.ctx.set_nu(net['c'] * np.exp(np.random.default_rng(1).normal(0, 0.01, net['n_tokens'])))\n"
                "print(json.dumps(min(1e6 * p.ctx.time_eval_kernel(_lib.TIME_TABLE, %d) for _ in range(%d))))" % ([ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")], reps, rounds))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CFMM_TABLE_WARM="0"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        out["table.stable4_1e5_cold_us"] = float(r.stdout.strip().splitlines()[-1])

    def utility_table():
        # a solve through the generic two-launch first-order iteration (utilities with CFMM_ULOG entries): device us per evaluation
        # (5e4 pools of every reference kind / 1000 tokens, a log utility that prices the tokens within ~10 % of the market: the instance of
        #  tests/test_gpu_utility.py)
        net = synthetic.config("C3", scale=0.05, seed=2)
        n = net["n_tokens"]
        rng = np.random.default_rng(0)
        hold = np.exp(rng.normal(3, 0.5, n)) / net["prices"]
        u = cfmm.LogUtility(hold * net["prices"] * np.exp(rng.normal(0, 0.1, n)), hold)
        prob = cfmm.Problem.from_network(net, utility=u)
        prob.solve(tol=1e-6, method="lbfgs")
        assert prob.status == "optimal", prob.status

        def iteration():
            prob.solve(tol=1e-6, method="lbfgs")
            return 1e6 * prob.stats["device_seconds"] / prob.stats["evals"]
        out["ulog.5e4_pools_us_per_evaluation"] = best(iteration)
        out["ulog.5e4_pools_evaluations_per_solve"] = float(prob.stats["evals"])          # (a count, not a time: recorded, never over budget by itself)
        prob.close()

    def sweep():
        # the reference's own sweep (two-asset.py:34-100, all five pools, 50 points) through cfmm_solve_sweep: LIBRARY wall time per sweep, us
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle import instances as I
        from helpers import problem_of
        p = problem_of(I.two_asset(0.0))
        utils = [cfmm.Swap([t, 0, 0], 2) for t in I.two_asset_sweep()]
        p.solve_many(utils, tol=1e-8)
        out["sweep.two_asset_50_points_library_us"] = best(lambda: 1e6 * p.solve_many(utils, tol=1e-8)[0]["stats"]["wall_seconds"])
        p.close()

    # (C4: 1e7 constant-product pools -- the large-set instantiations: compact mirror of ids and fees, 256-pool tiles; C4x4: 4e7, streamed)
    jobs = {"C3": lambda: first_order("C3", batch=True), "C4shard": lambda: first_order("C4shard"), "C2": lambda: first_order("C2"),
            "C4": lambda: first_order("C4"), "C5": second_order, "C4x4": lambda: first_order("C4x4"), "table": table, "ulog": utility_table,
            "sweep": sweep}
    for name, job in jobs.items():
        if only is None or name in only:
            job()
    return out


def compare(measured, budget):
    """-> list of (key, measured, allowed) over budget"""
    return [(k, v, budget["allowed_us"][k]) for k, v in measured.items() if k in budget["allowed_us"] and not k.endswith("_per_solve") and v > budget["allowed_us"][k]]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    m = measure(args.rounds, only=args.only)
    if args.write:
        json.dump({"note": "measured on MI355X by tools/kernel_budget.py --write (minimum over rounds of mean launch time, us); allowed = measured x %.2f (launch chains: x %.2f)" % (MARGIN, MARGIN_CHAIN),
                   "measured_us": {k: round(v, 3) for k, v in m.items()},
                   "allowed_us": {k: round(v * margin(k), 3) for k, v in m.items()}}, open(BUDGET, "w"), indent=1)
        print(json.dumps(m))
    else:
        b = json.load(open(BUDGET))
        over = compare(m, b)
        print(json.dumps({"measured_us": m, "over_budget": over}))
        sys.exit(1 if over else 0)
