"""GPU suite (-m gpu): the utility table (csrc/lbfgs_rules.hpp: utility_term; include/cfmm.h CFMM_ULOG / CFMM_UQUAD; SURVEY 8(f)
rank 4, "utilities beyond linear-plus-box") -- separable concave utilities through the generic two-launch first-order iteration:
the device against the C twin that restates the same iteration (oracle/cfmm_oracle.c: oracle_step), small instances against the
PRIMAL program with the same utility (oracle/primal_scipy.py, SLSQP), the certificates recomputed in NumPy, and the refusals of
the paths that do not take such utilities.  Tolerances: objectives 2e-6 relative (both sides at 1e-6 / 1e-7 certificates)."""
import numpy as np
import pytest

import cfmm
from cfmm import synthetic, _lib
from oracle import primal_scipy

pytestmark = pytest.mark.gpu


def _utility_value(u, psi):
    lin, lg, qd = u.ctype <= 2, u.ctype == cfmm.ULOG, u.ctype == cfmm.UQUAD
    return float(u.c[lin] @ psi[lin] + u.c[lg] @ np.log(psi[lg] + u.h[lg]) + u.c[qd] @ psi[qd] - 0.5 * psi[qd] ** 2 @ (1.0 / u.h[qd]))


def _conjugate(u, nu):
    lin, lg, qd = u.ctype <= 2, u.ctype == cfmm.ULOG, u.ctype == cfmm.UQUAD
    return float(((nu - u.c) * u.h)[lin].sum() + (u.c[lg] * np.log(u.c[lg] / nu[lg]) - u.c[lg] + nu[lg] * u.h[lg]).sum()
                 + (0.5 * u.h[qd] * (u.c[qd] - nu[qd]) ** 2).sum())


def _utilities(net, seed=0):
    n = net["n_tokens"]
    rng = np.random.default_rng(seed)
    pi = net["prices"]
    hold = np.exp(rng.normal(3, 0.5, n)) / pi                          # ~20 units of value per token ...
    log_u = cfmm.LogUtility(hold * pi * np.exp(rng.normal(0, 0.1, n)), hold)      # ... at weights that price them within 10 % of the market
    quad_u = cfmm.QuadraticUtility(net["c"], np.where(rng.random(n) < 0.7, np.exp(rng.normal(4, 0.5, n)) / pi ** 2, np.inf))
    ct = np.where(rng.random(n) < 0.4, cfmm.ULOG, np.where(rng.random(n) < 0.5, cfmm.UQUAD, cfmm.GE)).astype(np.int32)
    c = np.where(ct == cfmm.ULOG, log_u.c, net["c"])
    h = np.where(ct == cfmm.ULOG, log_u.h, np.where(ct == cfmm.UQUAD, np.exp(rng.normal(4, 0.5, n)) / pi ** 2, 0.0))
    return dict(log=log_u, quadratic=quad_u, mixed=cfmm.Utility(c, h, ct))


@pytest.mark.parametrize("which", ["log", "quadratic", "mixed"])
def test_utility_table_device_against_the_twin_and_its_own_certificates(oracle_lib, which):
    # 5e4 pools of every reference kind over 1000 tokens; utilities that price the tokens within ~10 % of the market (30-65
    # evaluations; an agent whose log utility values its holdings 20x off the market makes the first-order iteration crawl --
    # hundreds to thousands of evaluations: DESIGN.md)
    net = synthetic.config("C3", scale=0.05, seed=2)
    n = net["n_tokens"]
    u = _utilities(net)[which]
    p = cfmm.Problem.from_network(net, utility=u)
    v = p.solve(tol=1e-7, max_evals=8000)
    assert p.status == "optimal" and p.gap <= 1e-7 and p.infeas <= 1e-7, (p.status, p.gap, p.infeas, p.stats["evals"])
    # the certificates, recomputed: primal value U(psi), dual value ubar(nu) + nu'psi, their difference
    assert abs(v - _utility_value(u, p.psi)) <= 1e-9 * max(1.0, abs(v))
    dual = _conjugate(u, p.nu) + float(p.nu @ p.psi)
    assert abs(p.dual_value - dual) <= 1e-9 * max(1.0, abs(dual)) and -1e-9 * abs(dual) <= dual - v <= 2e-7 * max(1.0, abs(dual))
    lg = u.ctype == cfmm.ULOG
    assert np.all(p.psi[lg] + u.h[lg] > 0)
    # the C twin runs the same iteration on the CPU: the same optimum
    o = oracle_lib.Oracle(n, threads=4); o.add_network(net); o.set_utility(u.c, u.h, u.ctype)
    r = o.solve(cfmm.start_prices(net, u), tol=1e-7, max_evals=8000)
    assert r["status"] == 1
    assert abs(v - r["primal_value"]) <= 2e-6 * max(1.0, abs(v)) and abs(p.dual_value - r["dual_value"]) <= 2e-6 * max(1.0, abs(v))
    assert p.stats["evals"] <= 2 * r["evals"] + 16
    p.close()


def test_utility_table_small_instances_against_the_scipy_primal():
    rng = np.random.default_rng(5)
    n, m = 6, 14
    pi = np.exp(rng.normal(0, 0.5, n))
    L, R, G, K, W = [], [], [], [], []
    for i in range(m):
        k = 2 if i < 10 else 3
        l = rng.choice(n, k, replace=False)
        L.append(l); R.append(np.exp(rng.normal(3, 0.5)) / pi[l] * np.exp(rng.normal(0, 0.05, k))); G.append(0.997)
        K.append("geomean"); W.append(np.full(k, 1.0 / k))
    for u in (cfmm.LogUtility([1.0, 2.0, 0.5, 1.5, 1.0, 0.7], [5.0, 2.0, 8.0, 3.0, 4.0, 6.0]),
              cfmm.QuadraticUtility(pi * np.exp(rng.normal(0, 0.05, n)), [20.0, 30.0, np.inf, 25.0, 40.0, np.inf])):
        p = cfmm.Problem(n, L, R, G, K, W, utility=u)
        v = p.solve(tol=1e-9)
        assert p.status == "optimal", p.status
        r = primal_scipy.solve_primal(dict(n_tokens=n, local_indices=L, reserves=R, fees=G, kinds=K, weights=W, c=u.c, h=u.h, ctype=u.ctype))
        assert abs(v - r["value"]) <= 2e-8 * max(1.0, abs(v)) and np.abs(p.psi - r["psi"]).max() <= 2e-3 * max(1.0, np.abs(r["psi"]).max())
        p.close()


def test_utility_table_refusals():
    """what does not take such utilities says so: the batched solves, price ties, bad parameters"""
    net = synthetic.config("C3", scale=0.01, seed=1)
    n = net["n_tokens"]
    u = _utilities(net)["log"]
    p = cfmm.Problem.from_network(net, utility=u)
    assert p.solve(tol=1e-6) is not None and p.status == "optimal"          # (the default method takes the first-order path)
    res = p.solve_many([u, _utilities(net, seed=3)["log"]], tol=1e-6)        # not batched (one at a time, on clones): still solved
    assert all(r["status"] == "optimal" for r in res)
    ctx = _lib.Context(n)
    with pytest.raises(_lib.CfmmError, match="needs c > 0"):
        ctx.set_utility(np.zeros(n), np.ones(n), np.full(n, cfmm.ULOG, dtype=np.int32))
    with pytest.raises(_lib.CfmmError, match="needs h > 0"):
        ctx.set_utility(np.ones(n), np.zeros(n), np.full(n, cfmm.UQUAD, dtype=np.int32))
    ctx.set_utility(u.c, u.h, u.ctype)
    with pytest.raises(_lib.CfmmError, match="no price ties"):
        ctx.set_ties(np.arange(n, dtype=np.int32) // 2, np.zeros(n))
    ctx.close(); p.close()


@pytest.mark.parametrize("which", ["log", "mixed"])
def test_utility_table_through_the_second_order_path(which):
    """the barrier-smoothed Newton iteration keeps the utility on the host (cfmm_hip.hip: solve_newton): a table entry adds its
    conjugate, nu (psi - P*) to the gradient and nu^2 ubar'' to the Hessian's diagonal -- the same optimum as the first-order
    path, in a dozen steps where an agent far from the market costs the first-order iteration hundreds of evaluations"""
    net = synthetic.config("C3", scale=0.03, seed=4)
    n = net["n_tokens"]
    rng = np.random.default_rng(9)
    hold = np.exp(rng.normal(3, 0.5, n)) / net["prices"]
    far = cfmm.LogUtility(np.exp(rng.normal(0, 0.3, n)), hold)                 # values its holdings ~20x off the market
    u = far if which == "log" else _utilities(net)["mixed"]
    p = cfmm.Problem.from_network(net, utility=u)
    v2 = p.solve(method="newton", tol=1e-7)
    assert p.status == "optimal" and p.gap <= 1e-7 and p.infeas <= 1e-7, (p.status, p.gap, p.infeas)
    steps, psi2 = p.stats["newton_steps"], p.psi.copy()
    assert abs(v2 - _utility_value(u, psi2)) <= 1e-9 * max(1.0, abs(v2))
    v1 = p.solve(method="lbfgs", tol=1e-7, max_evals=8000)
    assert p.status == "optimal", (p.status, p.stats["evals"])
    assert abs(v1 - v2) <= 1e-6 * max(1.0, abs(v1)) and steps <= 40
    p.close()



def test_utility_table_over_a_network_with_k_asset_constant_sum_pools():
    """round 5 refused a utility-table entry over ANY network holding constant-sum pools of the K-asset table, for every
    method -- what cannot be had there is the tie loop.  The second-order path smooths these pools with its own barrier and needs no
    ties; `auto` takes a plain first-order leg and falls back on it.  Both end on certificates, at the same optimum."""
    net = synthetic.make_network(120, m_cp2=4000, m_w2=800, m_gk_sum=120, gk_sizes=(3, 4), seed=5)
    n = net["n_tokens"]
    rng = np.random.default_rng(1)
    hold = np.exp(rng.normal(3, 0.5, n)) / net["prices"]
    u = cfmm.LogUtility(hold * net["prices"] * np.exp(rng.normal(0, 0.05, n)), hold)
    p = cfmm.Problem.from_network(net, utility=u)
    v2 = p.solve(tol=1e-7, method="newton")
    assert p.status == "optimal" and p.gap <= 1e-7 and p.infeas <= 1e-7, (p.status, p.gap, p.infeas)
    assert np.all(p.psi + u.h > 0)
    v = p.solve(tol=1e-7, max_evals=6000)                         # auto: first order without ties, the second-order path if that ends uncertified
    assert p.status == "optimal" and abs(v - v2) <= 2e-6 * max(1.0, abs(v2)), (p.status, v, v2)
    p.close()
