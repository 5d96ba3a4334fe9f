"""TEST INFRASTRUCTURE (oracle side) -- the three problem instances the reference ships.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.

Each function returns a plain dict in the vocabulary of the reference scripts
(`local_indices`, `reserves`, `fees`) plus two things the reference encodes only
implicitly by *which constraint line it writes* for a pool:

  kinds[i]   : "geomean" (weighted geometric mean; Uniswap v2 is the equal-weight
               2-asset case) or "sum" (constant sum with non-negative reserves)
  weights[i] : the `p=` argument of the geo-mean (None -> equal weights)

and the utility:

  utility = {"type": "arbitrage",  "c": market_value}            max c'psi, psi >= 0
          | {"type": "liquidate",  "h": current_assets, "t": t}  max psi_t, psi_k = -h_k (k != t)
          | {"type": "swap",       "h": current_assets, "t": t}  max psi_t, psi + h >= 0

Problem data restated from (PARITY UNPINNED by the reference: it holds no expected outputs):
  arbitrage()   <- /root/reference/arbitrage.py:5-36   (pools), :57,:77 (utility)
  liquidation() <- /root/reference/liquidation.py:5-36 (pools), :57,:77-80 (utility)
  two_asset(t)  <- /root/reference/two-asset.py:7-32   (pools), :41-45,:66,:86 (utility)
"""
import numpy as np


def arbitrage():
    # arbitrage.py:65 writes the Balancer RHS as the *unweighted* geo_mean of the old
    # reserves; this equals the weighted one only because all reserves are 4.
    return dict(
        name="arbitrage",
        n_tokens=4,
        local_indices=[[0, 1, 2, 3], [0, 1], [1, 2], [2, 3], [2, 3]],
        reserves=[[4., 4., 4., 4.], [10., 1.], [1., 5.], [40., 50.], [10., 10.]],
        fees=[.998, .997, .997, .997, .999],
        kinds=["geomean", "geomean", "geomean", "geomean", "sum"],
        weights=[[4., 3., 2., 1.], None, None, None, None],
        utility=dict(type="arbitrage", c=[1.5, 10., 2., 3.]),
    )


def liquidation():
    return dict(
        name="liquidation",
        n_tokens=5,
        local_indices=[[0, 1, 2, 3, 4], [0, 1], [2, 3], [3, 4], [3, 4]],
        reserves=[[4., 4., 4., 4., 4.], [10., 1.], [1., 5.], [40., 50.], [10., 10.]],
        fees=[.998, .997, .997, .997, .999],
        kinds=["geomean", "geomean", "geomean", "geomean", "sum"],
        weights=[[5., 4., 3., 2., 1.], None, None, None, None],
        # current_assets[4] = 10 is never used by liquidation.py:77-80 (psi[4] is unconstrained)
        utility=dict(type="liquidate", h=[2., 1., 3., 5., 10.], t=4),
    )


def two_asset(t):
    """`t` is the amount of token 0 offered (two-asset.py:34,41-45); the sweep is
    np.linspace(0, 50) -> 50 points."""
    return dict(
        name="two_asset",
        n_tokens=3,
        local_indices=[[0, 1, 2], [0, 1], [1, 2], [0, 2], [0, 2]],
        reserves=[[3., .2, 1.], [10., 1.], [1., 10.], [20., 50.], [10., 10.]],
        fees=[.98, .99, .96, .97, .99],
        kinds=["geomean", "geomean", "geomean", "geomean", "sum"],
        weights=[[3., 2., 1.], None, None, None, None],
        utility=dict(type="swap", h=[float(t), 0., 0.], t=2),
    )


def two_asset_sweep():
    return np.linspace(0, 50)


def normalise(inst):
    """Lists -> numpy, weights normalised to sum 1, utility -> the unified (c, h, ctype)
    form used everywhere else:

        maximise c'psi  s.t.  psi_k + h_k >= 0 (ctype 0, 'GE')
                              psi_k + h_k  = 0 (ctype 1, 'EQ')
                              psi_k free       (ctype 2, 'FREE')
    """
    n = inst["n_tokens"]
    out = dict(inst)
    out["local_indices"] = [np.asarray(l, dtype=np.int32) for l in inst["local_indices"]]
    out["reserves"] = [np.asarray(r, dtype=np.float64) for r in inst["reserves"]]
    out["fees"] = np.asarray(inst["fees"], dtype=np.float64)
    ws = []
    for l, w in zip(out["local_indices"], inst["weights"]):
        w = np.ones(len(l)) if w is None else np.asarray(w, dtype=np.float64)
        ws.append(w / w.sum())
    out["weights"] = ws
    u = inst["utility"]
    c = np.zeros(n)
    h = np.zeros(n)
    ctype = np.zeros(n, dtype=np.int32)
    if u["type"] == "arbitrage":
        c[:] = u["c"]
    elif u["type"] == "liquidate":
        c[u["t"]] = 1.0
        h[:] = u["h"]
        h[u["t"]] = 0.0
        ctype[:] = 1
        ctype[u["t"]] = 2
    elif u["type"] == "swap":
        c[u["t"]] = 1.0
        h[:] = u["h"]
    else:
        raise ValueError(u["type"])
    out["c"], out["h"], out["ctype"] = c, h, ctype
    return out
