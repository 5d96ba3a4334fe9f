"""GPU suite (-m gpu), second-order path (CFMM_METHOD_NEWTON): the barrier-smoothed evaluation against its NumPy
restatement (oracle/barrier_np.py), the dense Cholesky against LAPACK, and the solves it drives against the SciPy
primal, the first-order solver and -- at BASELINE config 5's full size -- size-independent properties.
Tolerances: smoothed values / psi / Hessian 1e-9 relative to their own scale (fp64, different inner solvers:
bisection + Newton on the CPU, a safeguarded barrier-exact iteration on the device); objectives 1e-6 relative."""
import os

import numpy as np
import pytest

import cfmm
from cfmm import synthetic, _lib
from oracle import barrier_np
from helpers import problem_of, random_instance, normalise_with_params

pytestmark = pytest.mark.gpu


def _basket(net, seed=1, k=10):
    n = net["n_tokens"]
    rng = np.random.default_rng(seed)
    h = np.zeros(n); idx = rng.choice(n, min(k, n - 1), replace=False)
    h[idx] = np.exp(rng.normal(2, 0.5, len(idx))) / net["prices"][idx] * 10
    tgt = int(rng.integers(0, n)); h[tgt] = 0
    return h, tgt


def _mixed_network(seed=0, n=60, m=1500):
    """every two-asset family, constant-sum pools between equal-priced tokens of a peg group"""
    net = synthetic.make_network(n, m_cp2=m, m_w2=m, m_curve2=m, seed=seed)
    rng = np.random.default_rng(seed + 99)
    ms = m // 2
    ia = rng.integers(0, n, ms); ib = (ia // 4) * 4 + (ia % 4 + rng.integers(1, 4, ms)) % 4
    ib = np.minimum(ib, n - 1); ib = np.where(ib == ia, (ia // 4) * 4, ib)
    keep = ia != ib
    ia, ib = ia[keep].astype(np.int32), ib[keep].astype(np.int32)
    L = np.exp(rng.normal(np.log(1e3), 1.0, len(ia)))
    net["sum2"] = dict(Ra=L / net["prices"][ia], Rb=L / net["prices"][ia] * np.exp(rng.normal(0, 0.05, len(ia))),
                       fee=np.full(len(ia), 0.999), ia=ia, ib=ib)
    return net


@pytest.mark.parametrize("n", [37, 256, 1000])
def test_dense_cholesky_against_lapack(n):
    rng = np.random.default_rng(n)
    M = rng.normal(size=(n, max(4, n // 8)))
    A = M @ M.T + np.diag(rng.uniform(0.5, 2.0, n))
    b = rng.normal(size=n)
    ctx = _lib.Context(n)
    x, info = ctx.debug_cholesky(A, b)
    xr = np.linalg.solve(A, b)
    assert info == 0
    assert np.abs(x - xr).max() <= 1e-10 * np.abs(xr).max()
    # a NEW right-hand side through the same factor and its inverse factor (the chord step's two matrix-vector products)
    b2 = rng.normal(size=n)
    x2 = ctx.debug_cholesky_apply(b2)
    assert np.abs(x2 - np.linalg.solve(A, b2)).max() <= 1e-10 * np.abs(np.linalg.solve(A, b2)).max()
    # badly scaled but positive definite: the log-price Hessian spans many orders of magnitude
    s = np.exp(rng.normal(0, 6, n))
    A2 = A * s[:, None] * s[None, :]
    x2, info2 = ctx.debug_cholesky(A2, b)
    assert info2 == 0
    assert np.abs(x2 * s - np.linalg.solve(A, b / s)).max() <= 1e-8 * np.abs(np.linalg.solve(A, b / s)).max()
    # indefinite: flagged, not silently factored
    _, info3 = ctx.debug_cholesky(A - 1e3 * np.eye(n), b)
    assert info3 != 0
    ctx.close()


def _with_env(monkeypatch, **env):
    """a context created under tuning knobs that libcfmm_hip.so reads once per context (cfmm_create)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)


@pytest.mark.parametrize("n", [33, 100, 200, 1000, 2050])
def test_cholesky_pairs_of_block_columns_against_one_per_launch(n, monkeypatch):
    """chol2.hpp (two block columns per launch, side products on the fp64 matrix pipe) against chol.hpp (one per launch):
    solution, the chord step's apply through the inverse factor, the flag of an indefinite matrix"""
    rng = np.random.default_rng(n)
    M = rng.normal(size=(n, n)) / np.sqrt(n)
    A = M @ M.T + np.diag(rng.uniform(0.5, 2.0, n))
    b, b2 = rng.normal(size=n), rng.normal(size=n)
    out = {}
    for mode in ("pairs", "single"):
        _with_env(monkeypatch, CFMM_CHOL=mode)
        ctx = _lib.Context(n)
        x, info = ctx.debug_cholesky(A, b)
        x2 = ctx.debug_cholesky_apply(b2)
        _, info_bad = ctx.debug_cholesky(A - 1e3 * np.eye(n), b)
        ctx.close()
        assert info == 0 and info_bad != 0
        out[mode] = (x, x2)
    xr, xr2 = np.linalg.solve(A, b), np.linalg.solve(A, b2)
    for mode in out:
        assert np.abs(out[mode][0] - xr).max() <= 1e-10 * np.abs(xr).max(), mode
        assert np.abs(out[mode][1] - xr2).max() <= 1e-10 * np.abs(xr2).max(), mode
    assert np.abs(out["pairs"][0] - out["single"][0]).max() <= 1e-11 * np.abs(xr).max()


def test_second_order_hand_off_and_factorisation_variants_agree(monkeypatch):
    """the lean host <-> device hand-off (handoff.hpp: one launch per group of small copies, a flag in pinned memory instead of a
    stream synchronisation) against the hipMemcpyAsync / hipMemsetAsync path, and both factorisations: the same solve"""
    net = synthetic.config("C5", scale=0.1)
    n = net["n_tokens"]
    rng = np.random.default_rng(2)
    h = np.zeros(n); idx = rng.choice(n, 6, replace=False); h[idx] = rng.uniform(1.0, 20.0, 6)
    t = int(np.setdiff1d(np.arange(n), idx)[0])
    res = {}
    for io, ch in (("lean", "pairs"), ("blit", "pairs"), ("lean", "single")):
        _with_env(monkeypatch, CFMM_NEWTON_IO=io, CFMM_CHOL=ch)
        p = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t))
        v = p.solve(method="newton", tol=1e-7)
        assert p.status == "optimal" and p.gap <= 1e-7 and p.infeas <= 1e-7, (io, ch, p.status)
        res[(io, ch)] = (v, p.psi.copy(), p.stats["newton_steps"])
        p.close()
    v0, psi0, steps0 = res[("lean", "pairs")]
    vb, psib, stepsb = res[("blit", "pairs")]
    # (the hand-off moves bytes, not numbers: what differs is the order of the evaluation's fp64 atomics, run to run)
    assert abs(vb - v0) <= 1e-9 * abs(v0) and np.abs(psib - psi0).max() <= 1e-6 * np.abs(psi0).max() and stepsb == steps0
    vs, psis, _ = res[("lean", "single")]
    assert abs(vs - v0) <= 2e-7 * abs(v0) and np.abs(psis - psi0).max() <= 1e-5 * np.abs(psi0).max()


@pytest.mark.parametrize("mu", [1e-1, 1e-5, 1e-10])
def test_smoothed_evaluation_matches_numpy_restatement(mu):
    net = _mixed_network()
    n = net["n_tokens"]
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    ctx = p._ensure_ctx()
    rng = np.random.default_rng(5)
    nu = net["prices"] * np.exp(rng.normal(0, 0.03, n))
    val, tr, psi, H = ctx.eval_smooth(nu, mu, want_hessian=True)
    o = barrier_np.smooth_eval(net, nu, mu, hessian=True)
    assert abs(val - o["value"]) <= 1e-10 * max(1.0, abs(o["value"]))
    assert abs(tr - o["trade"]) <= 1e-10 * max(1.0, abs(o["trade"]))
    assert np.abs(psi - o["psi"]).max() <= 1e-10 * np.abs(o["psi"]).max()
    Hl, Ho = np.tril(H), np.tril(o["H"])
    assert np.abs(Hl - Ho).max() <= 1e-8 * np.abs(Ho).max()
    assert np.all(np.triu(H, 1) == 0.0)
    # the smoothed trade value never exceeds the exact optimum, and is within mu per barrier term of it
    arb, _ = ctx.eval_dual(nu)
    nbar = 2 * sum(len(net[k]["Ra"]) for k in ("cp2", "w2", "curve2")) + 4 * len(net["sum2"]["Ra"])
    assert -1e-9 * abs(arb) <= arb - tr <= mu * nbar * (1 + 1e-9) + 1e-9 * abs(arb)
    p.close()


def test_smoothed_tenders_stay_inside_every_trading_set():
    """the primal point of a second-order solve: Delta, Lambda > 0 and phi(R + gamma Delta - Lambda) >= phi(R)"""
    net = _mixed_network(seed=3)
    h, t = _basket(net)
    p = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t))
    p.solve(method="newton")
    assert p.status == "optimal" and p.stats["method"] == _lib.METHODS["newton"] and p.stats["barrier_mu"] > 0
    psi = np.zeros(net["n_tokens"])
    for key in ("cp2", "w2", "curve2", "sum2"):
        b = net[key]
        d, l = p.bucket_trades(key)
        assert d.min() > 0 and l.min() > 0
        xa = b["Ra"] + b["fee"] * d[0] - l[0]; xb = b["Rb"] + b["fee"] * d[1] - l[1]
        assert xa.min() > 0 and xb.min() > 0
        if key == "cp2":
            before, after = np.log(b["Ra"]) + np.log(b["Rb"]), np.log(xa) + np.log(xb)
        elif key == "w2":
            before = b["wa"] * np.log(b["Ra"]) + (1 - b["wa"]) * np.log(b["Rb"]); after = b["wa"] * np.log(xa) + (1 - b["wa"]) * np.log(xb)
        elif key == "curve2":
            before = b["Ra"] + b["Rb"] - b["alpha"] / (b["Ra"] * b["Rb"]); after = xa + xb - b["alpha"] / (xa * xb)
        else:
            before, after = b["Ra"] + b["Rb"], xa + xb
        assert np.all(after >= before - 1e-9 * np.abs(before)), (key, float(((before - after) / np.abs(before)).max()))
        np.add.at(psi, b["ia"], l[0] - d[0]); np.add.at(psi, b["ib"], l[1] - d[1])
    assert np.abs(psi - p.psi).max() <= 1e-9 * np.abs(p.psi).max()       # psi IS the scatter-sum of the tenders
    assert p.dual_value >= p.value - 1e-9 * abs(p.value)
    p.close()


@pytest.mark.parametrize("seed", range(12))
def test_second_order_small_instances_vs_primal(seed):
    """random 14-pool instances over every pool family (seeds >= 6 include 3..5-asset Balancer pools) and the three
    utilities, against SciPy SLSQP on the primal exactly as the scripts state it"""
    from oracle.primal_scipy import solve_primal
    util = ["liquidate", "swap", "arbitrage"][seed % 3]
    inst = random_instance(400 + seed, n_tokens=6, n_pools=14, with_sum=True, with_curve=True, utility=util, two_asset_only=seed < 6)
    p = problem_of(inst)
    v = p.solve(tol=1e-8, method="newton")
    r = solve_primal(normalise_with_params(inst))
    if p.status != "optimal":                 # a token to sell that no pool lists: SLSQP fails on it too
        assert not r["success"], (seed, p.status, p.gap, p.infeas, v, r["value"], p.stats)
        return
    info = dict(seed=seed, status=p.status, value=v, dual=p.dual_value, gap=p.gap, infeas=p.infeas, slsqp=r["value"], slsqp_ok=r["success"],
                steps=p.stats.get("newton_steps"), evals=p.stats.get("evals"))
    assert p.gap <= 1e-7 and p.infeas <= 1e-7, info
    assert r["value"] <= p.dual_value + 2e-6 * max(1, abs(v)), info           # weak duality vs SLSQP's point
    if r["success"]:
        assert abs(v - r["value"]) <= 2e-6 * max(1, abs(v)), info
    p.close()


@pytest.mark.parametrize("util", ["arbitrage", "swap", "liquidate"])
def test_second_order_agrees_with_first_order_on_constant_product_pools(util):
    net = synthetic.config("C2")
    if util == "arbitrage":
        u = cfmm.Arbitrage(net["c"])
    else:
        h, t = _basket(net, seed=7, k=5)
        u = cfmm.Swap(h, t) if util == "swap" else cfmm.Liquidate(h, t)
    p = cfmm.Problem.from_network(net, utility=u)
    v1 = p.solve(method="lbfgs"); s1 = p.status; psi1 = p.psi.copy()
    v2 = p.solve(method="newton"); s2 = p.status
    assert s1 == "optimal" and s2 == "optimal"
    assert p.stats["method"] == _lib.METHODS["newton"] and p.stats["newton_steps"] > 0
    assert abs(v1 - v2) <= 2e-6 * max(1.0, abs(v1)), (v1, v2)
    assert np.abs(psi1 - p.psi).max() <= 2e-4 * max(np.abs(psi1).max(), np.abs(u.h).max())
    p.close()


def test_config5_full_size_second_order():
    """BASELINE config 5: 5e5 stableswap pools (+ 5e4 constant-product), 1000 tokens, basket liquidation.
    The first-order iteration does not reach its certificates here in thousands of evaluations."""
    net = synthetic.config("C5")
    h, t = _basket(net)
    p = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t))
    v = p.solve()                                        # auto picks the second-order method for this network
    st = p.stats
    assert p.status == "optimal" and st["method"] == _lib.METHODS["newton"]
    assert p.gap <= 1e-6 and p.infeas <= 1e-6
    assert st["newton_steps"] <= 60 and st["evals"] <= 300
    r = p.psi + h
    assert np.abs(np.delete(r, t)).max() <= 1e-6 * max(np.abs(p.psi).max(), h.max())       # the basket is sold, nothing else moves
    assert v > 0 and p.dual_value >= v * (1 - 1e-9) and (p.dual_value - v) <= 1e-6 * p.dual_value
    # same optimum from a different barrier schedule and a perturbed start (the certificates are self-contained)
    nu0 = cfmm.start_prices(net, p.utility) * np.exp(np.random.default_rng(0).normal(0, 0.05, net["n_tokens"]))
    ctx = p._ensure_ctx()
    ctx.set_utility(p.utility.c, p.utility.h, p.utility.ctype)
    st2 = ctx.solve(nu0, method="newton", barrier_shrink=0.5)
    assert st2["status"] == 1 and abs(st2["primal_value"] - v) <= 2e-6 * v
    # ... and from an INDEPENDENT solve of the same instance at full size: oracle/barrier_newton.py (NumPy smoothed evaluations,
    # LAPACK Cholesky, its own fixed barrier schedule; certificates from the C oracle's exact dual), run once on the CPU by
    # oracle/make_c5_fixture.py (half an hour) -- no SciPy primal reaches this size and the C oracle's first-order iteration is
    # still 0.4 % away after 2000 evaluations.  The instance is rebuilt from the same seeds here.
    import json
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c5_liquidation.json")))
    assert fx["pools"] == p.m and fx["tokens"] == net["n_tokens"] and fx["target"] == t
    assert fx["gap"] <= 1e-6 and fx["infeas"] <= 1e-6 and fx["primal_value"] <= fx["dual_value"] * (1 + 1e-12)
    for mine in (v, p.dual_value):                       # both brackets overlap: the two optima agree to the certificates' 1e-6
        assert abs(mine - fx["dual_value"]) <= 2e-6 * v and abs(mine - fx["primal_value"]) <= 2e-6 * v, (mine, fx["dual_value"], fx["primal_value"])
    # the exact dual evaluation AT the independent solver's final prices: HIP kernel against the C oracle's value in the fixture
    nu_fx = np.asarray(fx["nu"])
    f_hip, _ = p.eval_dual(nu_fx)
    u = p.utility
    assert abs(f_hip + float((nu_fx - u.c) @ u.h) - fx["dual_value"]) <= 1e-9 * fx["dual_value"]
    p.close()


def test_many_constant_sum_pools_both_paths_agree():
    """2000 constant-sum pools among 20000: the first-order path (host-side active-set loop over the kinks) and the
    second-order path (closed-form smoothed fills, no host loop) reach the same optimum"""
    rng = np.random.default_rng(11)
    n = 200
    net = synthetic.make_network(n, m_cp2=18000, seed=11)
    m = 2000
    ia = rng.integers(0, n, m); ib = (ia + rng.integers(1, n, m)) % n
    L = np.exp(rng.normal(np.log(1e3), 1.0, m)); pi = net["prices"]
    # a constant-sum pool quotes 1:1: express both reserves in units of equal value so that it sits near the market
    net["sum2"] = dict(Ra=L / pi[ia], Rb=L / pi[ib], fee=np.full(m, 0.999), ia=ia.astype(np.int32), ib=ib.astype(np.int32))
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    v2 = p.solve(method="newton")
    assert p.stats["method"] == _lib.METHODS["newton"]
    assert p.status == "optimal" and p.gap <= 1e-6 and p.infeas <= 1e-6
    assert v2 >= 0 and p.dual_value >= v2 - 1e-9 * abs(v2)
    d, l = p.bucket_trades("sum2")
    b = net["sum2"]
    assert d.min() > 0 and (b["Ra"] + b["fee"] * d[0] - l[0]).min() > 0 and (b["Rb"] + b["fee"] * d[1] - l[1]).min() > 0
    v1 = p.solve()                                       # auto: constant-sum pools alone stay first order
    assert p.stats["method"] == _lib.METHODS["lbfgs"] and p.status == "optimal"
    assert abs(v1 - v2) <= 2e-6 * abs(v1), (v1, v2)
    p.close()


def test_second_order_refuses_what_it_cannot_take():
    net = synthetic.config("C2", scale=0.1)
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    with pytest.raises(ValueError):
        p.solve(method="simplex")
    ctx = p._ensure_ctx()
    ctx.set_utility(net["c"])
    n = net["n_tokens"]
    grp = np.arange(n, dtype=np.int32); grp[1] = 0; grp[2:] -= 1         # tokens 0 and 1 tied
    ctx.set_ties(grp, np.zeros(n))
    with pytest.raises(cfmm.CfmmError, match="ties"):
        ctx.solve(net["prices"], method="newton")
    with pytest.raises(cfmm.CfmmError):
        ctx.eval_smooth(net["prices"], 1e-3)
    ctx.set_ties(None, None)
    with pytest.raises(cfmm.CfmmError):
        ctx.eval_smooth(net["prices"], 0.0)                              # mu must be positive
    assert ctx.solve(net["prices"], method="newton")["status"] == 1
    p.close()


def test_second_order_token_limit_is_refused_not_launched():
    """Between ~5.8k and ~6.1k tokens the evaluation's LDS tiles still fit but the Hessian instantiation of the smoothed kernel
    (psi tile + diagonal + pair cache) does not: the second-order path must say so (CFMM_E_UNSUPPORTED) instead of failing
    in its launch, AUTO must stay first order, and the first-order path must still solve."""
    net = synthetic.make_network(6000, m_cp2=40000, seed=3)
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    ctx = p._ensure_ctx()
    ctx.set_utility(net["c"])
    with pytest.raises(cfmm.CfmmError, match="5792 tokens"):
        ctx.solve(net["c"], method="newton")
    with pytest.raises(cfmm.CfmmError, match="5792 tokens"):
        ctx.eval_smooth(net["prices"], 1e-3)
    p.solve(tol=1e-6)
    assert p.status == "optimal" and p.stats["method"] == _lib.METHODS["lbfgs"]
    p.close()


@pytest.mark.parametrize("mu", [1e-2, 1e-8])
def test_smoothed_evaluation_with_k_asset_pools(mu):
    """k-asset geo-mean pools ride along unsmoothed: exact solution, exact generalised Hessian"""
    net = synthetic.make_network(40, m_cp2=300, m_w2=200, m_gn=400, m_curve2=200, seed=8)
    n = net["n_tokens"]
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    ctx = p._ensure_ctx()
    nu = net["prices"] * np.exp(np.random.default_rng(2).normal(0, 0.05, n))
    val, tr, psi, H = ctx.eval_smooth(nu, mu, want_hessian=True)
    o = barrier_np.smooth_eval(net, nu, mu, hessian=True)
    assert abs(val - o["value"]) <= 1e-10 * max(1.0, abs(o["value"]))
    assert abs(tr - o["trade"]) <= 1e-10 * max(1.0, abs(o["trade"]))
    assert np.abs(psi - o["psi"]).max() <= 1e-10 * np.abs(o["psi"]).max()
    assert np.abs(np.tril(H) - np.tril(o["H"])).max() <= 1e-8 * np.abs(o["H"]).max()
    p.close()


def test_shipped_instances_through_the_second_order_path():
    """arbitrage.py / liquidation.py / two-asset.py as shipped (Balancer + Uniswap-v2 + a constant-sum pool each, the
    constant-sum pool partially filled at the optimum) against the golden optima and tenders
    (tests/golden/shipped_instances.json), at a tolerance the first-order path needs its host-side kink recovery for"""
    from helpers import golden, shipped_cases
    g = golden()
    for name, inst in shipped_cases():
        p = problem_of(inst)
        v = p.solve(tol=1e-9, method="newton")
        want = g[name]["kkt"]             # the 50-digit KKT solution (oracle/kkt_mp.py)
        assert p.stats["method"] == _lib.METHODS["newton"], name
        assert p.status == "optimal" and p.gap <= 1e-9 and p.infeas <= 1e-9, (name, p.status, p.gap, p.infeas)
        assert abs(v - want["value"]) <= 1e-7 * max(1.0, abs(want["value"])), (name, v, want["value"])
        # the tenders handed out after a second-order solve are the barrier-smoothed interior point: both directions
        # of a pool are (minutely) open, so it is the NET tender that is compared -- 1e-6 absolute, the bar
        assert np.abs(p.psi - np.asarray(want["psi"])).max() <= 1e-6, name
        for i, y in enumerate(want["y"]):
            yi = np.asarray(p.lambdas[i]) - np.asarray(p.deltas[i])
            assert np.abs(yi - np.asarray(y)).max() <= 1e-6, (name, i)
        p.close()


def test_partially_filled_constant_sum_pools_reach_machine_feasibility():
    """the fill of a constant-sum pool sitting on its kink is set through a price difference far below the fp64
    resolution of log nu: the last Newton steps are carried in the low-order log-prices (smooth.hpp: apply_slo)"""
    net = _mixed_network(seed=3)
    h, t = _basket(net)
    p = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t))
    p.solve(method="newton", tol=1e-8)
    assert p.status == "optimal" and p.stats["status"] == 1
    assert p.gap <= 1e-8 and p.infeas <= 1e-8
    p.close()


def test_second_order_on_the_mixed_config3_network():
    """C3 at 5 % (Uniswap-v2 + 2-asset Balancer + 3..8-asset Balancer pools): both methods, same optimum"""
    net = synthetic.config("C3", scale=0.05)
    p = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
    v1 = p.solve(method="lbfgs"); s1 = p.status
    v2 = p.solve(method="newton"); s2 = p.status
    assert s1 == "optimal" and s2 == "optimal" and p.stats["method"] == _lib.METHODS["newton"]
    assert abs(v1 - v2) <= 2e-6 * max(1.0, abs(v1)), (v1, v2)
    p.close()


def test_second_order_pool_sharded_path_with_a_one_rank_communicator():
    """the pool-sharded second-order control flow on ONE GPU: RCCL all-reduce of [psi | value | trade], of the dense
    Hessian and of the barrier count, on the library's stream -- a communicator of one rank runs exactly what every
    rank of an 8-GPU job runs; same optimum as the unsharded solve"""
    import subprocess, sys, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, os, json
sys.path[:0] = [%r, %r, %r]
import numpy as np
import torch, torch.distributed as dist
import cfmm
from cfmm import synthetic
from test_gpu_newton import _basket
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
net = synthetic.config("C5", scale=0.02)
h, t = _basket(net)
u = cfmm.Liquidate(h, t)
p = cfmm.distributed.sharded_problem(net, u, dist=dist, device=0)
v = p.solve(method="newton")
q = cfmm.Problem.from_network(net, utility=u)
w = q.solve(method="newton")
print(json.dumps(dict(v=v, w=w, status=p.status, method=p.stats["method"], steps=p.stats["newton_steps"], steps1=q.stats["newton_steps"],
                      ranks=p.stats["n_ranks"], gap=p.gap, infeas=p.infeas)))
dist.destroy_process_group()
''' % (root, os.path.join(root, "cfmm-routing-code_amd"), os.path.join(root, "tests"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["status"] == "optimal" and out["ranks"] == 1 and out["method"] == _lib.METHODS["newton"]
    assert out["gap"] <= 1e-6 and out["infeas"] <= 1e-6
    assert abs(out["v"] - out["w"]) <= 1e-6 * abs(out["w"])


def test_second_order_after_pools_are_uploaded_again():
    """the per-direction warm starts are sized by the bucket: re-uploading a larger bucket must not reuse the old ones"""
    small = synthetic.make_network(40, m_cp2=200, m_curve2=2000, seed=0)
    big = synthetic.make_network(40, m_cp2=2000, m_curve2=20000, seed=1)
    big["prices"] = small["prices"]
    n = small["n_tokens"]
    h, t = _basket(small)
    u = cfmm.Liquidate(h, t)
    ctx = _lib.Context(n)
    for net in (small, big, small):
        for key, kind in (("cp2", 0), ("curve2", 3)):
            b = net[key]
            ctx.upload_pools2(kind, b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"], b.get("alpha"))
        ctx.set_utility(u.c, u.h, u.ctype)
        st = ctx.solve(cfmm.start_prices(net, u), method="newton")
        assert st["status"] == 1 and st["gap"] <= 1e-6 and st["infeas"] <= 1e-6, (len(net["curve2"]["Ra"]), st)
    ctx.close()


def test_second_order_reports_an_unsellable_token_as_infeasible():
    """a token that must be traded away (liquidation) but that no pool lists: its price collapses; the iteration stops
    early instead of spending its whole budget, and the problem is reported infeasible (the reference's cvxpy would
    set prob.status = 'infeasible')"""
    net = synthetic.make_network(7, m_cp2=60, m_curve2=60, seed=5)
    net["n_tokens"] = 8                                   # token 7 exists but no pool touches it
    net["prices"] = np.append(net["prices"], 1.0); net["c"] = np.append(net["c"], 1.0)
    h = np.zeros(8); h[7] = 3.0; h[2] = 1.0
    p = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, 0))
    p.solve(method="newton")
    assert p.status == "infeasible"
    assert p.stats["newton_steps"] < 150
    p.close()


def test_c_abi_auto_method_choice_and_fallback():
    """cfmm_solve with opts.method = CFMM_METHOD_AUTO (what a raw C-ABI caller gets): second order straight away from
    4096 stableswap pools on, first order below that, and a first-order run that ends without its certificates is
    handed on to the second-order method"""
    net = synthetic.config("C5", scale=0.02)                 # 10 000 stableswap pools
    h, t = _basket(net)
    u = cfmm.Liquidate(h, t)
    p = cfmm.Problem.from_network(net, utility=u)
    ctx = p._ensure_ctx(); ctx.set_utility(u.c, u.h, u.ctype)
    nu0 = cfmm.start_prices(net, u)
    st = ctx.solve(nu0)                                       # default opts: method 0
    assert st["method"] == _lib.METHODS["newton"] and st["status"] == 1
    p.close()
    small = synthetic.make_network(40, m_cp2=200, m_curve2=2000, seed=0)
    h, t = _basket(small)
    u = cfmm.Liquidate(h, t)
    q = cfmm.Problem.from_network(small, utility=u)
    ctx = q._ensure_ctx(); ctx.set_utility(u.c, u.h, u.ctype)
    nu0 = cfmm.start_prices(small, u)
    st = ctx.solve(nu0, max_evals=2000)
    assert st["status"] == 1                                  # whichever method finished it
    st = ctx.solve(nu0, max_evals=12)                         # first order cannot finish in 12 evaluations -> handed on
    assert st["method"] == _lib.METHODS["newton"] and st["evals"] > 12
    q.close()


def test_second_order_warm_start_over_a_basket_sweep():
    """the parametric use of two-asset.py:34-100 at scale: one resident pool set, the basket scaled up and down; a
    warm-started second-order solve (prices and barrier weight continued) needs a fraction of the cold solve's steps
    and lands on the same optimum"""
    net = synthetic.config("C5", scale=0.2)
    h, t = _basket(net)
    p = cfmm.Problem.from_network(net, utility=cfmm.Liquidate(h, t))
    p.solve(method="newton")
    cold_steps = p.stats["newton_steps"]
    assert p.status == "optimal"
    for f in (1.05, 1.5, 0.5):
        p.set_utility(cfmm.Liquidate(h * f, t))
        vw = p.solve(method="newton", warm_start=True)
        warm_steps = p.stats["newton_steps"]
        assert p.status == "optimal" and p.gap <= 1e-6 and p.infeas <= 1e-6
        vc = p.solve(method="newton")
        assert p.status == "optimal"
        assert abs(vw - vc) <= 2e-6 * abs(vc), (f, vw, vc)
        # (steps include the cheap chord steps of round 4 -- no factorisation, ~90 us against ~550: 7 against 11 at f = 0.5 is
        #  5 factorisations against 9)
        assert warm_steps <= 0.7 * cold_steps, (f, warm_steps, cold_steps)
    p.close()


_SHARED_DEVICE_WORKER = r"""
import os, sys
root = sys.argv[1]
for q in (root, os.path.join(root, "cfmm-routing-code_amd"), os.path.join(root, "tests")):
    sys.path.insert(0, q)
import numpy as np, cfmm
from helpers import random_instance, problem_of
bad = []
for rep in range(3):
    for seed in (1171, 1262, 1463):
        rng = np.random.default_rng(seed)
        kw = dict(n_tokens=int(rng.integers(3, 9)), n_pools=int(rng.integers(4, 24)), with_sum=bool(seed % 2), with_curve=bool((seed // 2) % 2),
                  with_power=bool((seed // 4) % 3 == 0), utility=["arbitrage", "swap", "liquidate"][seed % 3])
        p = problem_of(random_instance(seed, **kw))
        v = p.solve(tol=1e-9)
        v2 = p.solve(tol=1e-8, method="newton")
        if not (p.status == "optimal" and abs(v2 - v) <= 1e-6 * max(1.0, abs(v))):
            bad.append((seed, p.status, v, v2, p.stats.get("numeric_error")))
        p.close()
print("BAD", bad)
"""


def test_second_order_solves_of_processes_sharing_the_device(tmp_path):
    """six processes run explicit second-order solves on ONE device at once.  Round 6's fuzz campaign (six fuzzers side by side) ended
    three instances with NaN directions that no single process reproduced: the pair Cholesky factors IN PLACE, workgroup 0 stored the
    factor over the diagonal region while a row workgroup of the same launch -- dispatched late on the shared device -- had not loaded it
    yet.  The row workgroups now count themselves in and workgroup 0 waits for the launch's count (csrc/chol2.hpp); before that fix this
    test failed in about one process of three (the instances are fuzz_small.py's seeds 1171, 1262, 1463, three times each)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(_SHARED_DEVICE_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), root], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for _ in range(6)]
    outs = [q.communicate(timeout=600)[0] for q in procs]
    for q, o in zip(procs, outs):
        assert q.returncode == 0, o[-2000:]
        assert "BAD []" in o, o[-2000:]
