"""the cvxpy-compatible shim (cfmm.cvx).  CPU part: the shim's algebra and pattern match, and the three
shipped programs through it with the C oracle standing in for the device; when /root/reference is present (this container,
not the GPU box) the REAL scripts are executed with nothing changed but what `import cvxpy` resolves to."""
import contextlib
import io
import os
import sys
import types

import numpy as np
import pytest

import cfmm
import cfmm.cvx as cp
from oracle import instances as I
from helpers import golden, shipped_cases
from oracle_ctx import OracleContext
import cvx_models

REF = "/root/reference"


@pytest.fixture
def oracle_device(oracle_lib):
    cp.CONTEXT_FACTORY = lambda n: OracleContext(n)
    yield
    cp.CONTEXT_FACTORY = None


def _check(name, prob, goal, net, tender, receive, tol_y):
    k = golden()[name]["kkt"]
    assert prob.status == cp.OPTIMAL
    assert abs(prob.value - k["value"]) <= 1e-8 * max(1.0, abs(k["value"]))
    assert abs(goal.value - prob.value) <= 1e-12 * max(1.0, abs(prob.value))
    assert np.abs(net.value - np.asarray(k["psi"])).max() <= tol_y
    for d, l, y in zip(tender, receive, k["y"]):
        assert np.all(d.value >= 0) and np.all(l.value >= 0)
        assert np.abs((l.value - d.value) - np.asarray(y)).max() <= tol_y


@pytest.mark.parametrize("name,inst", shipped_cases())
def test_shipped_programs_through_the_shim(oracle_device, name, inst):
    prob, goal, net, tender, receive = cvx_models.build(cp, inst)
    v = prob.solve(tol=1e-10)
    assert v == prob.value
    _check(name, prob, goal, net, tender, receive, 2e-8)


def _run_reference_script(fname):
    """exec the reference's own script text with `cvxpy` resolving to the shim (and the plotting modules, which need a
    LaTeX installation, to inert stand-ins): nothing else is changed"""
    src = open(os.path.join(REF, fname)).read()
    fake_plt = types.ModuleType("matplotlib.pyplot")
    for fn in ("plot", "legend", "xlabel", "ylabel", "ylim", "savefig", "figure", "close"):
        setattr(fake_plt, fn, lambda *a, **k: None)
    fake_mpl = types.ModuleType("matplotlib"); fake_mpl.pyplot = fake_plt
    fake_latexify = types.ModuleType("latexify"); fake_latexify.latexify = lambda *a, **k: None
    saved = {k: sys.modules.get(k) for k in ("cvxpy", "matplotlib", "matplotlib.pyplot", "latexify")}
    sys.modules.update({"cvxpy": cp, "matplotlib": fake_mpl, "matplotlib.pyplot": fake_plt, "latexify": fake_latexify})
    out = io.StringIO()
    ns = {"__name__": "__main__"}
    try:
        with contextlib.redirect_stdout(out):
            exec(compile(src, fname, "exec"), ns)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ns, out.getvalue()


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout does not travel to the GPU box")
def test_the_reference_scripts_run_unmodified(oracle_device):
    g = golden()
    ns, out = _run_reference_script("arbitrage.py")
    assert abs(ns["prob"].value - g["arbitrage"]["kkt"]["value"]) <= 1e-7
    assert out.strip().startswith("Total output value: 21.49980")
    ns, out = _run_reference_script("liquidation.py")
    assert abs(ns["psi"].value[4] - g["liquidation"]["kkt"]["value"]) <= 1e-7
    assert out.strip().startswith("Total liquidated value: 15.88301")
    ns, out = _run_reference_script("two-asset.py")           # the 50-point sweep, 50 solves
    u = ns["u_t"]
    for j in (0, 1, 10, 25, 49):
        assert abs(u[j] - g[f"two_asset_{j}"]["kkt"]["value"]) <= 1e-7
        for k in range(5):
            assert np.abs(ns["all_values"][k][:, j] - np.asarray(g[f"two_asset_{j}"]["kkt"]["y"][k])).max() <= 1e-6
    assert np.all(np.diff(u) > 0) and out.count("Total liquidated value:") == 50


def test_affine_algebra_matches_numpy():
    rng = np.random.default_rng(0)
    x, y = cp.Variable(3, nonneg=True), cp.Variable(2, nonneg=True)
    A, B, c = rng.normal(size=(4, 3)), rng.normal(size=(4, 2)), rng.normal(size=4)
    e = A @ x - 2.0 * (B @ y) + c
    f = (c @ e) + e[1] - cp.sum(e) / 2
    x._value, y._value = rng.random(3), rng.random(2)
    ev = A @ x._value - 2.0 * (B @ y._value) + c
    assert np.allclose(e.value, ev) and np.isclose(f.value, c @ ev + ev[1] - ev.sum() / 2)
    assert np.allclose((c + e).value, c + ev) and np.allclose((c - e).value, c - ev) and np.allclose((e * 3).value, 3 * ev)
    assert cp.geo_mean(np.array([4.0, 4.0, 4.0, 4.0]), p=np.array([4, 3, 2, 1])) == pytest.approx(4.0)
    assert cp.sum(np.array([1.0, 2.0])) == 3.0


def test_models_outside_the_routing_class_are_refused(oracle_device):
    inst = I.arbitrage()
    prob, goal, net, tender, receive = cvx_models.build(cp, inst)
    bad = cp.Problem(goal, prob.constraints + [2.0 * net[0] >= 1.0])            # not an entry of psi
    with pytest.raises(NotImplementedError):
        bad.solve()
    x = cp.Variable(2, nonneg=True)
    with pytest.raises(NotImplementedError):
        cp.Problem(cp.Maximize(cp.sum(x)), [x <= 1]).solve()                    # no pools at all
    # a geo-mean right-hand side that is not the pool's current invariant
    d, l = cp.Variable(2, nonneg=True), cp.Variable(2, nonneg=True)
    R = np.array([1.0, 2.0])
    with pytest.raises(NotImplementedError, match="current"):
        cp.Problem(cp.Maximize((l - d)[0]), [cp.geo_mean(R + 0.99 * d - l) >= 5.0, (l - d)[1] + 1 >= 0]).solve()


def _other_functions_instance(seed):
    """a small network in which the stableswap and the power-sum pool appear as cvxpy constraint lines"""
    from helpers import random_instance
    inst = random_instance(40 + seed, n_tokens=5, n_pools=9, with_sum=False, with_curve=True, with_power=True)
    assert "curve" in inst["kinds"] and "powersum" in inst["kinds"]
    return inst


@pytest.mark.parametrize("seed", range(2))
def test_stableswap_and_power_sum_constraint_lines_are_recognised(oracle_device, seed):
    """`cp.sum(x) - alpha*cp.inv_prod(x) >= ...` and `cp.sum(cp.power(x, q)) >= ...` (DCP-valid cvxpy for the library's two
    trading functions the reference does not ship) map onto the curve / power-sum buckets: same optimum as the primal
    SciPy model of the same program"""
    from helpers import normalise_with_params
    from oracle.primal_scipy import solve_primal
    inst = _other_functions_instance(seed)
    prob, goal, net, tender, receive = cvx_models.build(cp, inst)
    v = prob.solve(tol=1e-9)
    r = solve_primal(normalise_with_params(inst))
    assert prob.status == cp.OPTIMAL and abs(v - r["value"]) <= 2e-6 * max(1.0, abs(v))
    assert sorted(k for k in ("curve2", "pow2") if k in prob.routing.net) == ["curve2", "pow2"]
    # a right-hand side that is not the function's value at the current reserves is refused, as for geo_mean
    d, l = cp.Variable(2, nonneg=True), cp.Variable(2, nonneg=True)
    R = np.array([3.0, 4.0])
    x = R + 0.99 * d - l
    with pytest.raises(NotImplementedError, match="current"):
        cp.Problem(cp.Maximize((l - d)[0]), [cp.sum(cp.power(x, 0.5)) >= 1.0, (l - d)[1] + 1 >= 0]).solve()
    with pytest.raises(NotImplementedError, match="same x"):
        cp.Problem(cp.Maximize((l - d)[0]), [cp.sum(R + 0.98 * d - l) - 2.0 * cp.inv_prod(x) >= float(R.sum() - 2.0 / R.prod()), (l - d)[1] + 1 >= 0]).solve()


def _concave_model(which):
    """a routing program with a separable concave objective, written the way cvxpy takes it (DCP-valid), over the 14 geometric-mean
    pools / 6 tokens of the utility-table tests; returns the cvx problem, psi and the same utility as a cfmm.Utility"""
    from test_oracle import _small_geomean_instance
    pi, inst = _small_geomean_instance()
    n = inst["n_tokens"]
    A = [np.eye(n)[:, l] for l in inst["local_indices"]]
    D = [cp.Variable(len(l), nonneg=True) for l in inst["local_indices"]]
    L = [cp.Variable(len(l), nonneg=True) for l in inst["local_indices"]]
    psi = cp.sum([Ai @ (l - d) for Ai, d, l in zip(A, D, L)])
    cons = [cp.geo_mean(R + g * d - l, p=w) >= cp.geo_mean(R, p=w)
            for R, g, w, d, l in zip(inst["reserves"], inst["fees"], inst["weights"], D, L)]
    if which == "log":
        a, h = np.array([1.0, 2.0, 0.5, 1.5, 1.0, 0.7]), np.array([5.0, 2.0, 8.0, 3.0, 4.0, 6.0])
        obj = cp.Maximize(cp.sum(cp.multiply(a, cp.log(psi + h))))
        u = cfmm.LogUtility(a, h)
    elif which == "quadratic":
        c = pi * np.exp(np.random.default_rng(1).normal(0, 0.05, n))
        depth = np.array([20.0, 30.0, 15.0, 25.0, 40.0, 35.0])
        obj = cp.Maximize(c @ psi - cp.sum(cp.multiply(1.0 / (2.0 * depth), cp.square(psi))))
        u = cfmm.QuadraticUtility(c, depth)
    else:            # two log tokens, two quadratic ones, two of the reference's linear-arbitrage kind (arbitrage.py:57,77)
        c = np.array([1.0, 2.0, pi[2] * 1.03, pi[3] * 0.97, pi[4] * 1.02, pi[5]])
        h = np.array([5.0, 2.0, 30.0, 20.0, 0.0, 0.0])
        obj = cp.Maximize(c[:2] @ cp.log(psi[:2] + h[:2]) + c[2:] @ psi[2:] - cp.sum_squares(psi[2:4] / np.sqrt(2.0 * h[2:4])))
        cons = cons + [psi[4:] >= 0]
        u = cfmm.Utility(c, h, np.array([cfmm.ULOG, cfmm.ULOG, cfmm.UQUAD, cfmm.UQUAD, cfmm.GE, cfmm.GE], dtype=np.int32))
    return cp.Problem(obj, cons), psi, u, inst


@pytest.mark.parametrize("which", ["log", "quadratic", "mixed"])
def test_separable_concave_objectives_through_the_shim(oracle_device, which):
    """beyond the reference's linear objectives: `cp.sum(cp.multiply(a, cp.log(psi + h)))`, `c @ psi - cp.sum(cp.multiply(k,
    cp.square(psi)))`, `cp.sum_squares`, mixed with linear entries -- mapped onto the utility table (include/cfmm.h: CFMM_ULOG /
    CFMM_UQUAD) and solved; against the primal program with the same utility handed to SLSQP, and `.value` of the objective
    recomputed from the tenders"""
    from oracle.primal_scipy import solve_primal
    prob, psi, u, inst = _concave_model(which)
    v = prob.solve(tol=1e-9)
    assert prob.status == cp.OPTIMAL and v == prob.value
    got = prob.routing.utility
    r = solve_primal(dict(inst, c=u.c, h=u.h, ctype=u.ctype))
    assert abs(v - r["value"]) <= 2e-8 * max(1.0, abs(v))
    assert np.abs(psi.value - r["psi"]).max() <= 2e-3 * max(1.0, np.abs(r["psi"]).max())
    # the shim's token order is its own: the utility it built is the model's, permuted
    assert sorted(zip(got.c.round(12), got.h.round(12), got.ctype)) == sorted(zip(u.c.round(12), u.h.round(12), u.ctype))


def test_concave_terms_refuse_what_the_table_does_not_hold(oracle_device):
    prob, psi, u, inst = _concave_model("log")
    cons = prob.constraints
    with pytest.raises(ValueError, match="not concave"):
        cp.Maximize(-cp.sum(cp.log(psi + 1.0)))
    with pytest.raises(ValueError, match="not concave"):
        cp.Maximize(cp.sum(cp.square(psi)))
    with pytest.raises(ValueError, match="close it"):
        cp.Maximize(cp.log(psi + 1.0))
    with pytest.raises(NotImplementedError, match="no other constraint"):
        cp.Problem(cp.Maximize(cp.sum(cp.log(psi + 1.0))), cons + [psi[0] >= 0]).solve()
    with pytest.raises(NotImplementedError, match="h_j >= 0"):
        cp.Problem(cp.Maximize(cp.sum(cp.log(psi - 1.0))), cons).solve()
    with pytest.raises(NotImplementedError, match="linear term"):
        cp.Problem(cp.Maximize(cp.sum(cp.log(psi + 1.0)) + psi[0]), cons).solve()
    with pytest.raises(NotImplementedError, match="psi_j itself"):
        cp.Problem(cp.Maximize(np.ones(6) @ psi - cp.sum_squares(psi + 1.0)), cons).solve()
    with pytest.raises(NotImplementedError, match="entries of psi"):
        cp.Problem(cp.Maximize(cp.sum(cp.log(2.0 * psi + 1.0))), cons).solve()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["log", "quadratic", "mixed"])
def test_separable_concave_objectives_through_the_shim_on_the_gpu(which):
    from oracle.primal_scipy import solve_primal
    cp.CONTEXT_FACTORY = None
    prob, psi, u, inst = _concave_model(which)
    v = prob.solve(tol=1e-8)
    r = solve_primal(dict(inst, c=u.c, h=u.h, ctype=u.ctype))
    assert prob.status == cp.OPTIMAL and abs(v - r["value"]) <= 2e-7 * max(1.0, abs(v))
    assert np.abs(psi.value - r["psi"]).max() <= 2e-3 * max(1.0, np.abs(r["psi"]).max())
    prob.routing.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(2))
def test_stableswap_and_power_sum_constraint_lines_on_the_gpu(seed):
    from helpers import normalise_with_params
    from oracle.primal_scipy import solve_primal
    cp.CONTEXT_FACTORY = None
    inst = _other_functions_instance(seed)
    prob, goal, net, tender, receive = cvx_models.build(cp, inst)
    v = prob.solve(tol=1e-9)
    r = solve_primal(normalise_with_params(inst))
    assert prob.status == cp.OPTIMAL and abs(v - r["value"]) <= 2e-6 * max(1.0, abs(v))
    prob.routing.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,inst", shipped_cases())
def test_shipped_programs_through_the_shim_on_the_gpu(name, inst):
    cp.CONTEXT_FACTORY = None
    prob, goal, net, tender, receive = cvx_models.build(cp, inst)
    prob.solve(tol=1e-10)
    _check(name, prob, goal, net, tender, receive, 5e-8)
    prob.routing.close()


@pytest.mark.gpu
def test_the_sweep_through_the_shim_keeps_its_pools_resident(monkeypatch):
    """two-asset.py:40-100 states a NEW cp.Problem over the SAME five pools for each of its 50 amounts.  The shim recognises the pool
    set (token lists, reserves, fees, functions, weights) and keeps it uploaded: from the second point on a model only sends its
    utility and starts from the previous prices.  Same optimum as with the residency switched off (and as the KKT fixture), every
    pool's tenders included; the wall time of both ways printed"""
    import time
    cp.CONTEXT_FACTORY = None
    ts = I.two_asset_sweep()
    g = golden()
    runs = {}
    for resident in (2, 0):
        monkeypatch.setattr(cp, "RESIDENT_MAX", resident)
        cp._resident.clear()
        vals, ys, routes = [], [], set()
        t0 = time.perf_counter()
        for t in ts:
            prob, goal, net, tender, receive = cvx_models.build(cp, I.two_asset(float(t)))
            vals.append(prob.solve(tol=1e-10))
            assert prob.status == cp.OPTIMAL
            ys.append([r.value - d.value for d, r in zip(tender, receive)])
            routes.add(id(prob.routing))
            if not resident:
                prob.routing.close()
        runs[resident] = (np.array(vals), ys, time.perf_counter() - t0, len(routes))
    for p in list(cp._resident.values()):
        p.close()
    cp._resident.clear()
    v_res, y_res, dt_res, n_res = runs[2]
    v_off, y_off, dt_off, n_off = runs[0]
    print(f"two-asset.py through cfmm.cvx, 50 models: {1e3 * dt_res:.1f} ms with the pools resident, {1e3 * dt_off:.1f} ms re-uploading them")
    assert n_res == 1                                   # ONE resident problem served the 50 models (ids of the closed ones recycle: not counted)
    assert np.abs(v_res - v_off).max() <= 1e-8
    for j in (0, 1, 10, 25, 49):
        k = g[f"two_asset_{j}"]["kkt"]
        assert abs(v_res[j] - k["value"]) <= 1e-8 * max(1.0, abs(k["value"]))
        for a, b in zip(y_res[j], k["y"]):
            assert np.abs(a - np.asarray(b)).max() <= 5e-8
    for a, b in zip(y_res, y_off):
        for u, w in zip(a, b):
            assert np.abs(u - w).max() <= 1e-6
