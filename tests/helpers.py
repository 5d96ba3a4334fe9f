import json
import os

import numpy as np

import cfmm
from cfmm import synthetic as _syn
from oracle import instances as I

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shipped_instances.json")


def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def utility_of(inst):
    u = inst["utility"]
    if u["type"] == "arbitrage":
        return cfmm.Arbitrage(u["c"])
    if u["type"] == "liquidate":
        return cfmm.Liquidate(u["h"], u["t"])
    return cfmm.Swap(u["h"], u["t"])


def problem_of(inst, ctx=None):
    p = cfmm.Problem(inst["n_tokens"], inst["local_indices"], inst["reserves"], inst["fees"], inst["kinds"],
                     inst["weights"], params=inst.get("params"), utility=utility_of(inst))
    if ctx is not None:
        p.ctx = ctx
    return p


def shipped_cases():
    sweep = I.two_asset_sweep()
    cases = [("arbitrage", I.arbitrage()), ("liquidation", I.liquidation())]
    for j in (0, 1, 10, 25, 49):
        cases.append((f"two_asset_{j}", I.two_asset(sweep[j])))
    return cases


def random_instance(seed, n_tokens=6, n_pools=12, with_sum=True, with_curve=False, utility="arbitrage", two_asset_only=False, with_power=False):
    """small random instance in the reference's vocabulary, connected enough to be interesting"""
    rng = np.random.default_rng(seed)
    price = np.exp(rng.normal(0, 0.5, n_tokens))
    L, R, F, K, W, P = [], [], [], [], [], []
    for i in range(n_pools):
        r = rng.random()
        if r < 0.2 and n_tokens >= 3 and not two_asset_only:
            k = int(rng.integers(3, min(5, n_tokens) + 1))
            kind = "geomean"
        else:
            k = 2
            kind = "sum" if (with_sum and r > 0.9) else ("curve" if (with_curve and r > 0.75) else ("powersum" if (with_power and r > 0.5) else "geomean"))
        l = rng.choice(n_tokens, size=k, replace=False)
        val = np.exp(rng.normal(3, 1))
        if kind == "geomean":
            w = rng.integers(1, 5, size=k).astype(float)
            if k == 2 and rng.random() < 0.6:
                w = np.ones(2)
            res = val * (w / w.sum()) / price[l] * np.exp(rng.normal(0, 0.1, k))
            W.append(w); P.append(None)
        elif kind == "sum":
            res = val / price[l].mean() * np.exp(rng.normal(0, 0.1, k)); W.append(None); P.append(None)
        elif kind == "powersum":       # marginal price (Rb / Ra)^t: reserves that sit near the market
            t = float(rng.choice([0.25, 0.5, 0.7]))
            ra = val / price[l[0]]
            res = np.array([ra, ra * (price[l[0]] / price[l[1]]) ** (1.0 / t)]) * np.exp(rng.normal(0, 0.05, 2))
            W.append(None); P.append(t)
        else:
            res = val / price[l].mean() * np.exp(rng.normal(0, 0.05, k)); W.append(None)
            P.append(float(_syn.curve_alpha_from_A(res[0], res[1], 20.0)))
        L.append(l.tolist()); R.append(res.tolist()); F.append(float(rng.choice([0.997, 0.999, 0.99]))); K.append(kind)
    inst = dict(name=f"rand{seed}", n_tokens=n_tokens, local_indices=L, reserves=R, fees=F, kinds=K, weights=W, params=P)
    if utility == "arbitrage":
        inst["utility"] = dict(type="arbitrage", c=(price * np.exp(rng.normal(0, 0.05, n_tokens))).tolist())
    elif utility == "swap":
        h = np.zeros(n_tokens); h[0] = float(np.exp(rng.normal(1, 0.5)))
        inst["utility"] = dict(type="swap", h=h.tolist(), t=n_tokens - 1)
    else:
        h = np.exp(rng.normal(0, 0.5, n_tokens)); h[n_tokens - 1] = 0
        inst["utility"] = dict(type="liquidate", h=h.tolist(), t=n_tokens - 1)
    return inst


def normalise_with_params(inst):
    out = I.normalise(inst)
    out["params"] = inst.get("params", [None] * len(inst["local_indices"]))
    return out


def table_instance(seed):
    """small instance with K-asset table pools (n-asset stableswap over 2..5 tokens; 40 %: an n-asset constant-sum pool over tokens of
    clearly different value) among constant-product pools, one of the reference's three utilities by seed: tools/fuzz_table.py's generator"""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 9))
    price = np.exp(rng.normal(0, 0.05, n))
    for j in range(n):
        if rng.random() < 0.3:
            price[j] *= float(rng.choice([0.5, 2.0, 3.0]))
    L, R, F, K, W, P = [], [], [], [], [], []
    for _ in range(int(rng.integers(4, 10))):
        l = rng.choice(n, 2, replace=False); val = np.exp(rng.normal(4, 0.7))
        L.append(l.tolist()); R.append((val / price[l] * np.exp(rng.normal(0, 0.05, 2))).tolist()); F.append(float(rng.choice([0.997, 0.999, 0.99])))
        K.append("geomean"); W.append([1.0, 1.0]); P.append(None)
    for _ in range(int(rng.integers(1, 4))):
        k = int(rng.integers(2, min(5, n) + 1))
        l = rng.choice(n, k, replace=False); val = np.exp(rng.normal(4, 0.7))
        res = val / price[l] * np.exp(rng.normal(0, 0.04, k))
        L.append(l.tolist()); R.append(res.tolist()); F.append(float(rng.choice([0.999, 0.9995, 0.997]))); K.append("curve"); W.append(None)
        P.append(float(np.prod(res) * res.mean() / float(rng.choice([5.0, 40.0, 200.0]))))
    with_sum = rng.random() < 0.4
    if with_sum:
        k = int(rng.integers(3, 5))
        l = rng.choice(n, k, replace=False)
        L.append(l.tolist()); R.append((30.0 / price[l] * np.exp(rng.normal(0, 0.1, k))).tolist()); F.append(float(rng.choice([0.997, 0.99]))); K.append("sum"); W.append(None); P.append(None)
    ut = ["arbitrage", "swap", "liquidate"][seed % 3]
    if ut == "arbitrage":
        u = dict(type="arbitrage", c=(price * np.exp(rng.normal(0, 0.03, n))).tolist())
    elif ut == "swap":
        h = np.zeros(n); h[0] = float(np.exp(rng.normal(2, 0.5))); u = dict(type="swap", h=h.tolist(), t=n - 1)
    else:
        h = np.exp(rng.normal(0.5, 0.5, n)); h[n - 1] = 0.0; u = dict(type="liquidate", h=h.tolist(), t=n - 1)
    return dict(name=f"tbl{seed}", n_tokens=n, local_indices=L, reserves=R, fees=F, kinds=K, weights=W, params=P, utility=u), with_sum
