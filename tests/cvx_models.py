"""Test fixture: the three shipped programs written against the cvxpy modelling surface (`cp` is passed in, so the same
text runs on cfmm.cvx and -- where it is installed -- on cvxpy itself).  A restatement, not a copy: one builder covers
arbitrage.py:39-84, liquidation.py:39-87 and one point of two-asset.py:47-100, driven by the instance dicts of
oracle/instances.py; it goes through exactly the calls the scripts make (Variable(nonneg=True), A_i @ (L - D), cp.sum of a
list, geo_mean with and without p=, the constant-sum pair of constraints, Maximize, Problem.solve, .value)."""
import numpy as np


def build(cp, inst):
    n = inst["n_tokens"]
    pools = list(zip(inst["local_indices"], inst["reserves"], inst["fees"], inst["kinds"], inst["weights"]))
    select = []                                    # the dense local -> global matrices of arbitrage.py:42-48
    for idx, *_ in pools:
        S = np.zeros((n, len(idx)))
        S[np.asarray(idx), np.arange(len(idx))] = 1.0
        select.append(S)
    tender = [cp.Variable(len(idx), nonneg=True) for idx, *_ in pools]       # Delta_i
    receive = [cp.Variable(len(idx), nonneg=True) for idx, *_ in pools]      # Lambda_i
    net = cp.sum([S @ (lam - dlt) for S, dlt, lam in zip(select, tender, receive)])
    after = [np.asarray(R, float) + fee * dlt - lam for (_, R, fee, _, _), dlt, lam in zip(pools, tender, receive)]
    cons = []
    for i, ((idx, R, fee, kind, w), x) in enumerate(zip(pools, after)):
        R = np.asarray(R, float)
        if kind == "geomean" and w is not None:
            cons.append(cp.geo_mean(x, p=np.asarray(w, float)) >= cp.geo_mean(R, p=np.asarray(w, float)))
        elif kind == "geomean":
            cons.append(cp.geo_mean(x) >= cp.geo_mean(R))
        elif kind == "curve":          # (not in the reference: the same pattern for the library's stableswap pool, DCP-valid cvxpy)
            al = float(inst["params"][i])
            cons.append(cp.sum(x) - al * cp.inv_prod(x) >= float(np.sum(R) - al / np.prod(R)))
        elif kind == "powersum":       # (likewise: the generic bucket's power-sum pool)
            q = 1.0 - float(inst["params"][i])
            cons.append(cp.sum(cp.power(x, q)) >= float(np.sum(R ** q)))
        else:
            cons += [cp.sum(x) >= cp.sum(R), x >= 0]
    u = inst["utility"]
    if u["type"] == "arbitrage":
        goal = cp.Maximize(np.asarray(u["c"], float) @ net)
        cons.append(net >= 0)
    elif u["type"] == "liquidate":
        goal = cp.Maximize(net[u["t"]])
        cons += [net[k] + u["h"][k] == 0 for k in range(n) if k != u["t"]]
    else:
        goal = cp.Maximize(net[u["t"]])
        cons.append(net + np.asarray(u["h"], float) >= 0)
    return cp.Problem(goal, cons), goal, net, tender, receive
