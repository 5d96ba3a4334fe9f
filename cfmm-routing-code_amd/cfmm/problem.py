"""Host-side mirror of the reference's problem object and of its `cp.Problem(...).solve()` call.

The reference (/root/reference/arbitrage.py, liquidation.py, two-asset.py) has no functions or
classes: a "problem" is the module-level data `local_indices`, `reserves`, `fees`
(arbitrage.py:6-28), the constraint lines that pick a trading function per pool
(arbitrage.py:63-74), and a utility (arbitrage.py:57,77 / liquidation.py:57,77-80 /
two-asset.py:66,86); the result surface is `prob.value`, `psi.value`, `deltas[i].value`,
`lambdas[i].value` (arbitrage.py:84, two-asset.py:94-100).  `Problem` keeps exactly that
vocabulary and hands the work to libcfmm_hip.so (include/cfmm.h).  There is no CPU path here:
without the HIP extension and a gfx950 device `solve()` raises.

Host logic that stays in Python (tiny, O(n) or O(#constant-sum pools)):
  * packing pools into per-kind SoA buckets (the dense A_i of arbitrage.py:42-48 become int32 ids);
  * start prices for utilities that do not name a price for every token;
  * constant-sum pools on their kink: tying the two prices (a linear equality in log-price),
    re-solving on the device, and recovering the fill fraction (primal recovery).
"""
import os
import sys
import numpy as np

from . import _lib
from ._lib import GE, EQ, FREE, ULOG, UQUAD, POOL_CP2, POOL_W2, POOL_SUM2, POOL_CURVE2, POOL_POW2, MAX_POOL_SIZE, CfmmError

KIND2 = dict(cp2=POOL_CP2, w2=POOL_W2, sum2=POOL_SUM2, curve2=POOL_CURVE2, pow2=POOL_POW2)
# the name of a two-asset bucket's parameter column (None: the kind has none)
PARAM2 = dict(cp2=None, w2="wa", sum2=None, curve2="alpha", pow2="t")


# ------------------------------------------------------------------------------- utilities
class Utility:
    """maximise c'psi  s.t.  psi_k + h_k >= 0 (GE) | = 0 (EQ) | unconstrained (FREE) -- the reference's utilities (arbitrage.py:57,77;
    liquidation.py:57,77-80; two-asset.py:66,86).  Beyond them, per token, an entry of the utility table:
    ctype ULOG: + c_k log(psi_k + h_k);  UQUAD: + c_k psi_k - psi_k^2 / (2 h_k)   (include/cfmm.h; both outer iterations take them: the
    generic two-launch first-order iteration, and the second-order path -- which `method="auto"` falls back to when the first-order
    run ends without its certificates)."""

    def __init__(self, c, h=None, ctype=None):
        self.c = np.asarray(c, dtype=np.float64)
        n = len(self.c)
        self.h = np.zeros(n) if h is None else np.asarray(h, dtype=np.float64)
        self.ctype = np.zeros(n, dtype=np.int32) if ctype is None else np.asarray(ctype, dtype=np.int32)


def Arbitrage(market_value):
    """max market_value @ psi, psi >= 0                      (arbitrage.py:57,77)"""
    return Utility(market_value)


def Liquidate(current_assets, target):
    """max psi[target], psi[k] + current_assets[k] == 0, k != target   (liquidation.py:57,77-80)"""
    h = np.asarray(current_assets, dtype=np.float64).copy()
    n = len(h)
    c = np.zeros(n); c[target] = 1.0
    h[target] = 0.0
    ctype = np.full(n, EQ, dtype=np.int32); ctype[target] = FREE
    return Utility(c, h, ctype)


def LogUtility(weights, holdings):
    """max sum_k weights[k] log(psi_k + holdings[k]): a logarithmic (Kelly / Cobb-Douglas) utility of the post-trade holdings.
    Not in the reference (its objectives are linear); tokens with weight 0 are held at psi_k + holdings[k] >= 0 and worth nothing."""
    a = np.asarray(weights, dtype=np.float64)
    h = np.asarray(holdings, dtype=np.float64)
    ctype = np.where(a > 0, ULOG, GE).astype(np.int32)
    return Utility(a, h, ctype)


def QuadraticUtility(marginal_value, depth):
    """max sum_k marginal_value[k] psi_k - psi_k^2 / (2 depth[k]): a linear value with quadratic impact (an external order book of
    finite depth behind every token).  depth[k] = inf is the linear-arbitrage token (psi_k >= 0 at value c_k).
    marginal_value >= 0 (what cfmm_set_utility accepts for every entry kind: include/cfmm.h)."""
    c = np.asarray(marginal_value, dtype=np.float64)
    if np.any(c < 0):
        raise ValueError("QuadraticUtility: marginal_value must be >= 0")
    d = np.asarray(depth, dtype=np.float64)
    fin = np.isfinite(d)
    return Utility(c, np.where(fin, d, 0.0), np.where(fin, UQUAD, GE).astype(np.int32))


def Swap(current_assets, target):
    """max psi[target], psi + current_assets >= 0            (two-asset.py:66,86)"""
    h = np.asarray(current_assets, dtype=np.float64)
    c = np.zeros(len(h)); c[target] = 1.0
    return Utility(c, h)


# ------------------------------------------------------------------------------- packing
def pack(n_tokens, local_indices, reserves, fees, kinds=None, weights=None, params=None):
    """Ragged pool list (reference vocabulary) -> SoA buckets + `where[i] = (bucket, position)`.

    kinds[i]: "geomean" (default) | "sum" | "curve" | "powersum";  weights[i]: geo-mean exponents (None =
    equal, i.e. Uniswap v2 for two assets);  params[i]: alpha of a curve pool, the exponent t of a power-sum pool
    (x^(1-t) + y^(1-t): the generic bucket's first tenant, include/cfmm.h CFMM_POOL_POW2).
    "sum" and "curve" pools over 3..8 tokens go to the K-asset table's buckets (net["gk"][(kind, k)]: csrc/phik.hpp,
    include/cfmm.h CFMM_POOLK_*): `where` then holds (("stable" | "sum", k), position)."""
    m = len(local_indices)
    kinds = ["geomean"] * m if kinds is None else list(kinds)
    weights = [None] * m if weights is None else list(weights)
    params = [None] * m if params is None else list(params)
    rows = dict(cp2=[], w2=[], sum2=[], curve2=[], pow2=[])
    rows_n = {}
    rows_g = {}
    where = []
    for i in range(m):
        l = np.asarray(local_indices[i], dtype=np.int64)
        R = np.asarray(reserves[i], dtype=np.float64)
        k = len(l)
        if len(R) != k:
            raise ValueError(f"pool {i}: {k} indices but {len(R)} reserves")
        if len(set(l.tolist())) != k or l.min() < 0 or l.max() >= n_tokens:
            raise ValueError(f"pool {i}: token ids must be distinct and in [0, {n_tokens})")
        if not np.all(R > 0) or not (0 < fees[i] <= 1):
            raise ValueError(f"pool {i}: reserves must be > 0 and the fee in (0, 1]")
        kind = kinds[i]
        if kind == "geomean":
            w = np.ones(k) if weights[i] is None else np.asarray(weights[i], dtype=np.float64)
            if len(w) != k or not np.all(w > 0):
                raise ValueError(f"pool {i}: bad weights")
            w = w / w.sum()
            if k == 2:
                if w[0] == w[1]:
                    where.append(("cp2", len(rows["cp2"])))
                    rows["cp2"].append((R[0], R[1], fees[i], 0.0, l[0], l[1]))
                else:
                    where.append(("w2", len(rows["w2"])))
                    rows["w2"].append((R[0], R[1], fees[i], w[0], l[0], l[1]))
            elif 3 <= k <= MAX_POOL_SIZE:
                b = rows_n.setdefault(k, [])
                where.append((k, len(b)))
                b.append((l, R, w, fees[i]))
            else:
                raise ValueError(f"pool {i}: geo-mean pools hold 2..{MAX_POOL_SIZE} tokens, got {k}")
        elif kind == "sum":
            if k == 2:
                where.append(("sum2", len(rows["sum2"])))
                rows["sum2"].append((R[0], R[1], fees[i], 0.0, l[0], l[1]))
            elif 3 <= k <= MAX_POOL_SIZE:                  # arbitrage.py:73-74 over more than two tokens: the K-asset table
                b = rows_g.setdefault(("sum", k), [])
                where.append((("sum", k), len(b)))
                b.append((l, R, fees[i], 0.0))
            else:
                raise ValueError(f"pool {i}: constant-sum pools hold 2..{MAX_POOL_SIZE} tokens, got {k}")
        elif kind == "curve":
            if params[i] is None or not (float(params[i]) > 0):
                raise ValueError(f"pool {i}: curve pools need params[i] = alpha > 0")
            if k == 2:
                where.append(("curve2", len(rows["curve2"])))
                rows["curve2"].append((R[0], R[1], fees[i], float(params[i]), l[0], l[1]))
            elif 3 <= k <= MAX_POOL_SIZE:                  # sum x - alpha / prod x over more than two tokens: the K-asset table
                b = rows_g.setdefault(("stable", k), [])
                where.append((("stable", k), len(b)))
                b.append((l, R, fees[i], float(params[i])))
            else:
                raise ValueError(f"pool {i}: curve pools hold 2..{MAX_POOL_SIZE} tokens, got {k}")
        elif kind == "powersum":
            if k != 2 or params[i] is None or not (1e-3 <= float(params[i]) <= 0.999):
                raise ValueError(f"pool {i}: power-sum pools are two-asset and need params[i] = t in [0.001, 0.999]")
            where.append(("pow2", len(rows["pow2"])))
            rows["pow2"].append((R[0], R[1], fees[i], float(params[i]), l[0], l[1]))
        else:
            raise ValueError(f"pool {i}: unknown kind {kind!r}")
    net = dict(n_tokens=int(n_tokens))
    pname = PARAM2
    for key, rr in rows.items():
        if not rr:
            continue
        a = np.array(rr, dtype=np.float64)
        b = dict(Ra=a[:, 0].copy(), Rb=a[:, 1].copy(), fee=a[:, 2].copy(),
                 ia=a[:, 4].astype(np.int32), ib=a[:, 5].astype(np.int32))
        if pname[key]:
            b[pname[key]] = a[:, 3].copy()
        net[key] = b
    if rows_n:
        net["gn"] = {}
        for k, rr in rows_n.items():
            net["gn"][k] = dict(idx=np.array([r[0] for r in rr], dtype=np.int32).T.copy(),
                                R=np.array([r[1] for r in rr]).T.copy(),
                                w=np.array([r[2] for r in rr]).T.copy(),
                                fee=np.array([r[3] for r in rr], dtype=np.float64))
    if rows_g:
        net["gk"] = {}
        for key, rr in rows_g.items():
            net["gk"][key] = dict(idx=np.array([r[0] for r in rr], dtype=np.int32).T.copy(),
                                  R=np.array([r[1] for r in rr]).T.copy(),
                                  fee=np.array([r[2] for r in rr], dtype=np.float64),
                                  param=np.array([r[3] for r in rr], dtype=np.float64))
    return net, where


def network_pool_count(net):
    m = sum(len(net[k]["Ra"]) for k in KIND2 if k in net)
    m += sum(b["R"].shape[1] for b in net.get("gn", {}).values())
    m += sum(b["R"].shape[1] for b in net.get("gk", {}).values())
    return m


def shard_network(net, rank, world):
    """Pool-sharding: contiguous equal-count slices of every bucket; tokens,
    prices and the utility stay replicated."""
    out = {k: v for k, v in net.items() if k not in KIND2 and k not in ("gn", "gk") and not str(k).startswith("_")}      # (not the per-network caches: _price_relations, _potentials)

    def sl(m):
        lo = (m * rank) // world
        hi = (m * (rank + 1)) // world
        return slice(lo, hi)
    for key in KIND2:
        if key in net:
            b = net[key]; s = sl(len(b["Ra"]))
            out[key] = {c: v[s] for c, v in b.items()}
    if "gn" in net:
        out["gn"] = {}
        for k, b in net["gn"].items():
            s = sl(b["R"].shape[1])
            out["gn"][k] = dict(idx=b["idx"][:, s], R=b["R"][:, s], w=b["w"][:, s], fee=b["fee"][s])
    if "gk" in net:
        out["gk"] = {}
        for key, b in net["gk"].items():
            s = sl(b["R"].shape[1])
            out["gk"][key] = dict(idx=b["idx"][:, s], R=b["R"][:, s], fee=b["fee"][s], param=b["param"][s])
    return out


# ------------------------------------------------------------------------------- start prices
def _is_general(util):
    """does the utility hold entries of the utility table (ULOG / UQUAD)?  Kept with the utility object like the other facts derived
    from its arrays (Problem.set_utility drops them when the object is re-sent): three array scans less per solve"""
    g = getattr(util, "_general", None)
    if g is None:
        g = bool((util.ctype >= ULOG).any())
        try:
            util._general = g
        except AttributeError:
            pass
    return g


def start_prices(net, util):
    """Prices for every token: c where the utility names one, otherwise propagated through the
    pools' marginal prices at their current reserves (breadth-first, averaged in log space).
    The propagation depends on the pools and on c alone: its result is kept with the utility object (a re-solve with
    another basket h -- or the same one -- skips the ~2 ms walk over a 1000-token network)."""
    c = util.c
    if _is_general(util):           # the utility table: a token starts at the price at which it would not trade (u'(0))
        c = c.copy()
        lg, qd = util.ctype == ULOG, util.ctype == UQUAD
        with np.errstate(divide="ignore"):
            c[lg] = np.where(util.h[lg] > 0, util.c[lg] / np.where(util.h[lg] > 0, util.h[lg], 1.0), 0.0)
        c[qd] = util.c[qd]
        known = c > 0
        return c if known.all() else _propagate_prices(net, c, known)
    if getattr(util, "_all_priced", None) is None:       # (kept with the utility object: Problem.set_utility drops it when the object is re-sent)
        try:
            util._all_priced = bool((c > 0).all())
        except AttributeError:
            pass
    known = c > 0 if getattr(util, "_all_priced", None) is None else None
    if getattr(util, "_all_priced", None) or (known is not None and known.all()):
        return c.copy()
    if known is None:
        known = c > 0
    memo = getattr(util, "_start_memo", None)
    if memo is not None and memo[0] is net and np.array_equal(memo[1], c):
        return memo[2].copy()
    out = _propagate_prices(net, c, known)
    try:
        util._start_memo = (net, c.copy(), out.copy())
    except AttributeError:
        pass
    return out


def _price_relations(net):
    """(eu, ev, lr): log p_eu - log p_ev = lr, the pools' marginal prices at their current reserves -- a property of the
    network alone, built once per network (kept in the dict under a private key; shard_network builds new dicts)."""
    rel = net.get("_price_relations")
    if rel is not None:
        return rel
    n = net["n_tokens"]
    eu, ev, elr = [], [], []
    # a rough guess is all this has to be (the solvers start by repairing it): on large networks every bucket is
    # thinned to an evenly strided sample, ~64 price relations per token in all
    total = sum(len(net[k]["Ra"]) for k in KIND2 if k in net) + \
        sum((kk - 1) * b["R"].shape[1] for kk, b in net.get("gn", {}).items()) + \
        sum((key[1] - 1) * b["R"].shape[1] for key, b in net.get("gk", {}).items())
    stride = max(1, total // (64 * n))
    for key in KIND2:
        if key not in net:
            continue
        b = net[key]
        sl = slice(0, None, stride)
        Ra, Rb = b["Ra"][sl], b["Rb"][sl]
        if key == "cp2":
            lr = np.log(Rb / Ra)
        elif key == "w2":
            wa = b["wa"][sl]
            lr = np.log(wa * Rb / ((1 - wa) * Ra))
        elif key == "sum2":
            lr = np.zeros(len(Ra))
        elif key == "pow2":
            lr = b["t"][sl] * np.log(Rb / Ra)
        else:
            al = b["alpha"][sl]
            lr = np.log((1 + al / (Ra * Ra * Rb)) / (1 + al / (Ra * Rb * Rb)))
        eu.append(b["ia"][sl]); ev.append(b["ib"][sl]); elr.append(lr)
    for k, b in net.get("gn", {}).items():
        sl = slice(0, None, stride)
        for j in range(1, k):
            eu.append(b["idx"][j][sl]); ev.append(b["idx"][0][sl])
            elr.append(np.log(b["w"][j][sl] * b["R"][0][sl] / (b["w"][0][sl] * b["R"][j][sl])))
    # the K-asset table's pools (csrc/phik.hpp): a token the utility leaves unpriced and that hangs on the rest through such pools
    # alone would otherwise be a component of its own and start at price 1.  Stableswap: the marginal ratio
    # (1 + s / R_j) / (1 + s / R_0), s = alpha / prod R; constant sum: equal prices
    for (kind, k), b in net.get("gk", {}).items():
        sl = slice(0, None, stride)
        R = b["R"][:, sl]
        if kind == "stable":
            sR = b["param"][sl] / np.prod(R, axis=0)
            lm = np.log1p(sR[None, :] / R)
        else:
            lm = np.zeros_like(R)
        for j in range(1, k):
            eu.append(b["idx"][j][sl]); ev.append(b["idx"][0][sl]); elr.append(lm[j] - lm[0])
    if eu:
        rel = (np.concatenate(eu).astype(np.int64), np.concatenate(ev).astype(np.int64), np.concatenate(elr))
    else:
        rel = (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0))
    net["_price_relations"] = rel
    return rel


def _potentials(net):
    """(phi, comp): log-price potentials of the tokens, the least-squares fit  min sum (phi_u - phi_v - lr)^2  over the
    network's price relations (one free constant per connected component), and the component of every token.  A property of
    the pools alone: solved once per network (sparse conjugate gradients; a few ms at 1000 tokens), after which the start prices of
    ANY utility cost O(n).  (Round 3 walked the relations breadth-first from the utility's priced tokens on every new utility:
    2-3 ms of host time per config-5 solve, hidden behind a per-utility memo.)"""
    pot = net.get("_potentials")
    if pot is not None:
        return pot
    n = net["n_tokens"]
    eu, ev, lr = _price_relations(net)
    if not len(eu):
        pot = (np.zeros(n), np.arange(n))
        net["_potentials"] = pot
        return pot
    # connected components: min-label propagation with pointer jumping (O(log n) rounds of two scatter-mins)
    comp = np.arange(n)
    while True:
        new = comp.copy()
        np.minimum.at(new, eu, comp[ev])
        np.minimum.at(new, ev, comp[eu])
        new = new[new]
        if np.array_equal(new, comp):
            break
        comp = new
    comp = np.unique(comp, return_inverse=True)[1]
    deg = (np.bincount(eu, minlength=n) + np.bincount(ev, minlength=n)).astype(float)

    def adj(x):                                            # A x for the (multi-)graph's adjacency matrix: two gathers, two scatter-adds
        return np.bincount(eu, weights=x[ev], minlength=n) + np.bincount(ev, weights=x[eu], minlength=n)
    rhs = np.bincount(eu, weights=lr, minlength=n) - np.bincount(ev, weights=lr, minlength=n)
    # Jacobi-preconditioned conjugate gradients on the graph Laplacian L = D - A (singular by one constant per component; the
    # right-hand side is orthogonal to those, and the iteration started at zero never leaves their complement)
    dinv = 1.0 / np.maximum(deg, 1.0)
    phi = np.zeros(n)
    r = rhs.copy()
    z = dinv * r
    pdir = z.copy()
    rz = float(r @ z)
    stop = 1e-12 * max(float(np.abs(rhs).max()), 1e-300)
    for _ in range(4 * n + 100):
        if float(np.abs(r).max()) <= stop:
            break
        q = deg * pdir - adj(pdir)
        alpha = rz / float(pdir @ q)
        phi += alpha * pdir
        r -= alpha * q
        z = dinv * r
        rz_new = float(r @ z)
        pdir = z + (rz_new / rz) * pdir
        rz = rz_new
    nc = int(comp.max()) + 1                              # gauge: zero mean per component
    phi -= (np.bincount(comp, weights=phi, minlength=nc) / np.bincount(comp, minlength=nc))[comp]
    pot = (phi, comp)
    net["_potentials"] = pot
    return pot


def _propagate_prices(net, c, known):
    """unpriced tokens from the network's potentials (above), shifted per connected component to agree on average with the
    prices the utility names there; a component the utility prices nowhere gets 1"""
    n = net["n_tokens"]
    phi, comp = _potentials(net)
    nc = int(comp.max()) + 1
    lc = np.log(np.where(known, c, 1.0))
    cnt = np.bincount(comp[known], minlength=nc)
    off = np.bincount(comp[known], weights=(lc - phi)[known], minlength=nc) / np.maximum(cnt, 1)
    logp = np.where(cnt[comp] > 0, phi + off[comp], 0.0)
    return np.where(known, c, np.exp(logp))


# ------------------------------------------------------------------------------- ties (kinks)
class _Ties:
    """Weighted union-find over tokens: log nu_j = s[group(j)] + off[j]."""

    def __init__(self, n):
        self.parent = np.arange(n)
        self.off = np.zeros(n)           # log nu_j - log nu_root(j)

    def find(self, j):
        path = []
        while self.parent[j] != j:
            path.append(j); j = self.parent[j]
        root = j
        # compress, accumulating offsets from the top of the path down
        for node in reversed(path):
            p = self.parent[node]
            if p != root:
                self.off[node] += self.off[p]
            self.parent[node] = root
        return root

    def tie(self, a, b, delta):
        """impose log nu_a - log nu_b = delta; False if it contradicts existing ties"""
        ra, rb = self.find(a), self.find(b)
        if ra == rb:
            return abs((self.off[a] - self.off[b]) - delta) < 1e-12
        # attach ra under rb: off[ra] = log nu_ra - log nu_rb
        self.off[ra] = delta + self.off[b] - self.off[a]
        self.parent[ra] = rb
        return True

    def groups(self):
        n = len(self.parent)
        roots = np.array([self.find(j) for j in range(n)])
        uniq, grp = np.unique(roots, return_inverse=True)
        return grp.astype(np.int32), self.off.copy(), len(uniq)


# ------------------------------------------------------------------------------- pool-sharded agreement
class HostComm:
    """The host-side agreement a pool-sharded solve needs (cfmm.distributed): every decision that changes the
    number or order of device collectives -- start prices, which method runs, which constant-sum pools are tied --
    must be the SAME on every rank, so it is taken on global quantities exchanged here.  Tiny and off the hot path
    (a few n-vectors per solve); wraps an initialised torch.distributed (nccl on the GPUs, gloo in the CPU tests)."""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        import torch
        self._torch = torch
        self._dev = "cuda" if dist.get_backend() == "nccl" else "cpu"

    def _t(self, a):
        return self._torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy()).to(self._dev)

    def broadcast(self, a, src=0):
        t = self._t(a)
        self.dist.broadcast(t, src=src)
        return t.cpu().numpy()

    def allreduce_sum(self, a):
        t = self._t(a)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allgather(self, obj):
        box = [None] * self.world
        self.dist.all_gather_object(box, obj)
        return box

    def assert_identical(self, a, what):
        """every rank must hold bit-identical `a` (different prices on different ranks would make the all-reduced
        psi a sum of shards evaluated at different points: silently wrong, or a hang)"""
        a = np.ascontiguousarray(a, dtype=np.float64)
        t = self._t(np.concatenate([a, -a]))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        mx = t.cpu().numpy()
        if not (np.array_equal(mx[:len(a)], a) and np.array_equal(-mx[len(a):], a)):
            raise CfmmError(f"pool-sharded solve: {what} differ between ranks")


# ------------------------------------------------------------------------------- Problem
AUTO_NEWTON_MIN_STABLE = 4096      # include/cfmm.h: CFMM_AUTO_NEWTON_MIN_STABLE


def _drain_leg(code):
    """the leg a K-asset drain record flags, from the last field of its key: j itself, or 200 + 10 t + j for the record of the SAME leg paid
    for by another of the pool's cheapest tokens (leg t): Problem._split_payers"""
    return code if code < 100 else (code - 200) % 10


class Problem:
    """Drop-in for the reference's `prob = cp.Problem(obj, cons); prob.solve()`.

        p = Problem(n_tokens, local_indices, reserves, fees, kinds, weights, utility=Arbitrage(mv))
        p.solve()            # -> objective value (what arbitrage.py:84 prints)
        p.value, p.status, p.psi, p.deltas[i], p.lambdas[i], p.nu, p.gap, p.stats
    """

    def __init__(self, n_tokens, local_indices=None, reserves=None, fees=None, kinds=None, weights=None,
                 params=None, utility=None, device=0, network=None, deterministic=False):
        self.n = int(n_tokens)
        if network is not None:
            self.net, self.where = network, None
        else:
            self.net, self.where = pack(self.n, local_indices, reserves, fees, kinds, weights, params)
        self.m = network_pool_count(self.net)
        self.utility = utility
        self.device = device
        self.deterministic = bool(deterministic)      # bitwise-reproducible psi (include/cfmm.h: cfmm_set_deterministic)
        self.ctx = None
        self._uploaded = False
        self.value = None; self.status = None; self.psi = None; self.nu = None
        self.gap = None; self.infeas = None; self.dual_value = None; self.stats = None
        self._theta = {}
        self._trade_cache = None
        self._comm = None              # (n_ranks, rank) once the library's RCCL communicator is up
        self._host = None              # HostComm of a pool-sharded problem (cfmm.distributed.sharded_problem)
        self._dev_utility = None       # the Utility object whose data sits on the device
        self._dev_ties = False         # price ties / pool flags are set on the device

    @classmethod
    def from_network(cls, net, utility=None, device=0, deterministic=False):
        return cls(net["n_tokens"], utility=utility, device=device, network=net, deterministic=deterministic)

    def clone(self, utility=None):
        """A Problem over the SAME pools -- they stay where they are in HBM, nothing is uploaded again --
        with its own utility, prices and solver state (cfmm_clone).  Clones may be solved concurrently from
        different host threads: `solve_many`."""
        ctx = self._ensure_ctx()
        q = Problem(self.n, utility=utility if utility is not None else self.utility, device=self.device, network=self.net,
                    deterministic=self.deterministic)        # (cfmm_clone hands the mode on to the new context)
        q.where = self.where
        q.ctx = ctx.clone()
        q._uploaded = True
        return q

    def solve_many(self, utilities, concurrency=2, nu0s=None, warm_start=False, batch=None, **kw):
        """Solve the same pools under many utilities (the two-asset.py:34-100 sweep; independent baskets).  Returns one
        dict per utility, in order: value, status, psi, nu, gap, infeas, stats.

        Batched (the default wherever it applies: first-order method, no constant-sum / stableswap pools, one GPU):
        `batch` solves at a time (default: as many as the LDS tile takes, 8 up to ~1100 tokens) run in lock-step through
        cfmm_solve_batch -- every outer iteration reads every pool column once for all of them.  With warm_start the
        solves of one group start from the prices the same slot reached in the previous group.
        Otherwise (`batch=0`, or a network the batched path does not take): `concurrency` clones work through the list
        from as many host threads."""
        utilities = list(utilities)
        ctx = self._ensure_ctx()
        if batch is None and nu0s is None and self._sweep_applies(ctx, utilities, kw):
            # (the test above mirrors the library's own -- tile sizes, the one-workgroup path switched on -- and cannot know a build or
            #  an environment that differs, CFMM_TINY=0, another WT_LIGHT: a library that declines falls through to the paths below
            #  instead of failing a call they would serve; ADVICE r5.  warm_start and concurrency mean nothing to the one-call sweep.)
            try:
                return self._solve_sweep(utilities, **kw)
            except CfmmError as e:
                if getattr(e, "code", None) != _lib.E_UNSUPPORTED:
                    raise
        can_batch = (hasattr(ctx, "solve_batch") and "sum2" not in self.net and "curve2" not in self.net and "pow2" not in self.net and self._host is None
                     and not any(_is_general(x) for x in utilities)
                     and not self.deterministic and kw.get("method", "auto") in ("auto", "lbfgs"))
        if batch is None:
            batch = ctx.batch_capacity() if can_batch else 0
        if batch and not can_batch:
            raise ValueError("solve_many(batch=...): the batched path takes first-order solves of networks without constant-sum / "
                             "stableswap pools on one GPU")
        if batch:
            return self._solve_batched(utilities, int(batch), nu0s, warm_start, **kw)
        import threading
        workers = [self] + [self.clone() for _ in range(max(1, int(concurrency)) - 1)]
        results = [None] * len(utilities)
        errors = []
        lock = threading.Lock()
        nxt = [0]

        def run(p):
            try:
                first = True
                while True:
                    with lock:
                        i = nxt[0]; nxt[0] += 1
                    if i >= len(utilities):
                        return
                    p.set_utility(utilities[i])
                    p.solve(nu0=None if nu0s is None else nu0s[i], warm_start=warm_start and not first, **kw)
                    first = False
                    results[i] = p._result()
            except Exception as e:       # surfaced in the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=run, args=(p,)) for p in workers]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for p in workers[1:]:
            p.close()
        if errors:
            raise errors[0]
        return results

    # -- the reference's own sweep (two-asset.py:34-100): tiny networks, constant-sum pools included, one library call --------
    _WT = dict(cp2=128, sum2=128, w2=64)               # pools per wave-tile (csrc/kernels.hpp: wave_tile_pools)

    def _sweep_applies(self, ctx, utilities, kw):
        """cfmm_solve_sweep serves what ONE workgroup evaluates: <= 64 tokens, <= 64 wave-tiles, linear-box utilities, first-order
        method, one GPU (csrc/tiny.hpp)"""
        if not hasattr(ctx, "solve_sweep") or self._host is not None or self.deterministic or self.n > 64 or not utilities:
            return False
        if kw.get("method", "auto") not in ("auto", "lbfgs") or any(k in self.net for k in ("curve2", "pow2", "gk")):
            return False
        if set(kw) - {"tol", "max_evals", "memory", "method", "kink_tol", "max_rounds"}:
            return False
        tiles = sum(-(-len(self.net[k]["Ra"]) // wt) for k, wt in self._WT.items() if k in self.net)
        tiles += sum(-(-b["R"].shape[1] // (64 // k)) for k, b in self.net.get("gn", {}).items())
        return 1 <= tiles <= 64 and not any(_is_general(u) for u in utilities)

    def _trade_layout(self):
        """(key, legs, pools, offset) of every non-empty bucket in cfmm_solve_sweep's tender layout, and the doubles per point"""
        lay, off = [], 0
        for key in KIND2:                                # (dict order = kind order 0..4)
            if key in self.net and len(self.net[key]["Ra"]):
                m = len(self.net[key]["Ra"]); lay.append((key, 2, m, off)); off += 4 * m
        for k in sorted(self.net.get("gn", {})):
            m = self.net["gn"][k]["R"].shape[1]
            if m:
                lay.append((k, k, m, off)); off += 2 * k * m
        return lay, off

    def _solve_sweep(self, utilities, tol=1e-6, max_evals=2000, memory=0, method="auto", kink_tol=1e-3, max_rounds=6):
        ctx = self._ensure_ctx()
        n, B = self.n, len(utilities)
        C_ = np.stack([u.c for u in utilities]); H = np.stack([u.h for u in utilities]); CT = np.stack([u.ctype for u in utilities]).astype(np.int32)
        # start prices: a function of the pools and of c alone (problem.py: start_prices) -- the points of a sweep mostly share c
        memo, NU0 = {}, np.empty((B, n))
        for b, u in enumerate(utilities):
            key = u.c.tobytes()
            if key not in memo:
                memo[key] = start_prices(self.net, u)
            NU0[b] = memo[key]
        lay, T = self._trade_layout()
        s2 = self.net.get("sum2")
        nu, psi, theta, tsgn, trades, sts, rounds = ctx.solve_sweep(C_, H, CT, NU0, s2, T, kink_tol, max_rounds, tol=tol, max_evals=max_evals,
                                                                    memory=memory, method=_lib.METHODS["lbfgs"])
        # fills of the pools that ended tied on their kink: psi_total = psi + sum theta d, their tenders = theta x the full fill
        tied = np.isfinite(theta) if s2 is not None else None
        if s2 is not None and tied.any():
            m2 = len(s2["Ra"])
            ya = np.where(tsgn > 0, -s2["Rb"] / s2["fee"], s2["Ra"]) * np.where(tied, theta, 0.0)      # [B][m2]: the pool's first token
            yb = np.where(tsgn > 0, s2["Rb"], -s2["Ra"] / s2["fee"]) * np.where(tied, theta, 0.0)      #          its second
            psi = psi.copy()
            np.add.at(psi, (np.arange(B)[:, None], s2["ia"][None, :]), ya)
            np.add.at(psi, (np.arange(B)[:, None], s2["ib"][None, :]), yb)
        # certificates, as Problem._finish computes them, for all points at once
        r = psi + H
        value = (C_ * psi).sum(axis=1)
        cs = ((nu - C_) * r).sum(axis=1)
        dual = ((nu - C_) * H).sum(axis=1) + (nu * psi).sum(axis=1)
        gap = np.abs(cs) / np.maximum(1.0, np.abs(dual))
        viol = np.where(CT == GE, np.maximum(-r, 0.0), np.where(CT == EQ, np.abs(r), 0.0)).max(axis=1)
        scale = np.maximum(np.maximum(np.abs(psi).max(axis=1), np.abs(H).max(axis=1)), max(1e-12 * self._max_reserve(), 1e-300))
        infeas = viol / scale
        tolx = max(tol, 1e-12) * (1 + 1e-6) + 1e-15
        good = (gap <= tolx) & (infeas <= tolx)
        collapsed = ((CT == EQ) & (H > 0) & ~self._listed_tokens()[None, :]).any(axis=1)      # (a token that must be sold and that no pool lists: _finish)
        results = []
        for b in range(B):
            st = sts[b]
            status = "optimal" if good[b] else ("infeasible" if collapsed[b] else
                                                 ("inaccurate" if st["status"] == 1 else _lib.STATUS.get(st["status"], f"error {st['status']}")))
            st.update(rounds=int(rounds[b]), batch=B, pool_subproblems=st["evals"] * self.m)
            res = dict(value=float(value[b]), status=status, psi=psi[b], nu=nu[b], gap=float(gap[b]), infeas=float(infeas[b]),
                       dual_value=float(dual[b]), stats=st)
            if trades is not None:
                tr = {}
                for key, k, m, off in lay:
                    blk = trades[b, off:off + 2 * k * m].reshape(2, k, m)
                    tr[key] = (blk[0], blk[1])
                if s2 is not None and tied[b].any():
                    d, l = tr["sum2"]
                    for i in np.flatnonzero(tied[b]):
                        y = np.array([ya[b, i], yb[b, i]])
                        d[:, i] = np.maximum(-y, 0.0); l[:, i] = np.maximum(y, 0.0)
                res["trades"] = tr
            results.append(res)
        # a point the sweep did not bring to its certificates takes the one-at-a-time path, second-order fall-back included
        if method == "auto":
            for b in np.flatnonzero(~good):
                self.set_utility(utilities[b])
                self.solve(tol=tol, max_evals=max_evals, memory=memory, kink_tol=kink_tol, max_rounds=max_rounds)
                res = self._result()
                res["trades"] = {k: v for k, v in self._trades().items()}
                results[int(b)] = res
        return results

    def tenders_of(self, result):
        """(deltas, lambdas) per pool, in the order of the pool list, of one result of a swept solve_many (two-asset.py:94,98)"""
        if self.where is None:
            raise CfmmError("per-pool lists need a Problem built from pool lists")
        tr = result["trades"]
        return ([tr[key][0][:, pos].copy() for key, pos in self.where], [tr[key][1][:, pos].copy() for key, pos in self.where])

    def _result(self):
        return dict(value=self.value, status=self.status, psi=self.psi, nu=self.nu, gap=self.gap, infeas=self.infeas,
                    dual_value=self.dual_value, stats=self.stats)

    def _solve_batched(self, utilities, batch, nu0s, warm_start, tol=1e-6, max_evals=2000, memory=0, method="auto", **kw):
        if kw:
            raise TypeError(f"solve_many: unknown option(s) {sorted(kw)}")
        ctx = self._ensure_ctx()
        batch = max(1, min(batch, ctx.batch_capacity(), len(utilities) or 1))
        workers = getattr(self, "_batch_workers", None) or [self]
        while len(workers) < batch:
            workers.append(self.clone())
        self._batch_workers = workers          # (kept: the next sweep over these pools reuses the clones' state buffers)
        results = []
        for g0 in range(0, len(utilities), batch):
            group = utilities[g0:g0 + batch]
            ws = workers[:len(group)]
            starts = []
            for k, (p, u) in enumerate(zip(ws, group)):
                p.utility = u
                p.ctx.set_utility(u.c, u.h, u.ctype)
                p._dev_utility = u
                p._theta = {}; p._trade_cache = None; p._tol = tol
                given = None if nu0s is None else nu0s[g0 + k]
                if given is not None:
                    starts.append(given)
                elif warm_start and p.nu is not None:
                    starts.append(p.nu)
                else:
                    starts.append(start_prices(self.net, u))
            sts = ws[0].ctx.solve_batch([p.ctx for p in ws[1:]], starts, tol=tol, max_evals=max_evals, memory=memory,
                                        method=_lib.METHODS["lbfgs"])
            for p, st in zip(ws, sts):
                nu, psi = p.ctx.get_solution()
                total = dict(evals=st["evals"], iters=st["iters"], wall_seconds=st["wall_seconds"], device_seconds=st["device_seconds"],
                             rounds=1, batch=len(group))
                p._finish(st, nu, psi, total)
                results.append(p._result())
        return results

    # -- device plumbing ---------------------------------------------------------------------
    def _ensure_ctx(self):
        if self.ctx is None:
            self.ctx = _lib.Context(self.n, self.device)      # raises without HIP lib / gfx950
            if self.deterministic:
                self.ctx.set_deterministic(True)
        if not self._uploaded:
            for key, kind in KIND2.items():
                if key in self.net:
                    b = self.net[key]
                    param = b.get(PARAM2[key]) if PARAM2[key] else None
                    self.ctx.upload_pools2(kind, b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"], param)
            for k, b in self.net.get("gn", {}).items():
                self.ctx.upload_poolsN(b["idx"], b["R"], b["w"], b["fee"])
            for (kind, k), b in self.net.get("gk", {}).items():       # the K-asset table's buckets (csrc/phik.hpp)
                self.ctx.upload_poolsG(_lib.POOLK[kind], b["idx"], b["R"], b["fee"], b["param"] if kind == "stable" else None)
            self._uploaded = True
        return self.ctx

    def set_utility(self, utility):
        self.utility = utility       # (a new object, or the same object mutated: call this to re-send it)
        self._dev_utility = None
        for attr in ("_all_priced", "_plain", "_start_memo", "_general"):        # (what was derived from its arrays)
            try:
                setattr(utility, attr, None)
            except AttributeError:
                pass

    def _send_utility(self):
        """the device-side utility is re-sent only when it changed (each call is a synchronisation).  Everything that
        runs a solve on the raw context (Problem.solve, cfmm.distributed.attach_oneshot_checked) goes through here:
        cfmm_solve refuses a context that never received a utility (CFMM_E_STATE)."""
        if self.utility is None:
            raise ValueError("no utility set")
        ctx = self._ensure_ctx()
        u = self.utility
        if self._dev_utility is not u:
            if self._host:
                # the utility is replicated: every rank must hold the same one (checked when it is sent to the device, once
                # per utility object -- the host-side collectives of a solve are off its steady-state path)
                self._host.assert_identical(np.concatenate([u.c, u.h, u.ctype.astype(np.float64)]), "the utilities")
            ctx.set_utility(u.c, u.h, u.ctype)
            self._dev_utility = u

    def init_comm(self, n_ranks, rank, uid):
        """pool-sharding: this process holds one shard; see cfmm.distributed"""
        self._ensure_ctx().comm_init(n_ranks, rank, uid)
        self._comm = (n_ranks, rank)

    def eval_dual(self, nu, want_diag=False):
        return self._ensure_ctx().eval_dual(nu, want_diag)

    # -- solve -------------------------------------------------------------------------------
    def solve(self, tol=1e-6, nu0=None, max_evals=2000, memory=0, iters_per_graph=8, kink_tol=1e-3,
              max_rounds=6, warm_start=False, method="auto"):
        """`method`: "lbfgs" (first order, on-device), "newton" (barrier-smoothed second order: the remedy for
        stableswap / constant-sum pools at scale; k-asset pools enter it unsmoothed) or "auto" (second order when the
        network holds stableswap pools; otherwise first order, with the host-side active-set loop for constant-sum
        kinks and the second-order method as its fall-back)."""
        if self.utility is None:
            raise ValueError("no utility set")
        ctx = self._ensure_ctx()
        u = self.utility
        self._send_utility()
        if self._dev_ties:
            self._clear_ties(ctx)
        nu0_given = nu0
        host = self._host
        rank = host.rank if host else 0
        # global pool counts: the branches below change the sequence of device collectives, so a pool-sharded solve
        # takes them on what ALL ranks hold, never on its own shard
        cnt = getattr(self, "_global_counts", None)
        if cnt is None:                               # (the pools of a Problem never change: one collective per Problem, not per solve)
            gk = self.net.get("gk", {})
            cnt = np.array([(len(self.net["curve2"]["Ra"]) if "curve2" in self.net else 0) + sum(b["R"].shape[1] for (kd, _), b in gk.items() if kd == "stable"),
                            (len(self.net["sum2"]["Ra"]) if "sum2" in self.net else 0) + sum(b["R"].shape[1] for (kd, _), b in gk.items() if kd == "sum")],
                           dtype=np.float64)
            if host:
                cnt = host.allreduce_sum(cnt)
            self._global_counts = cnt
        n_stable, n_sum = int(cnt[0]), int(cnt[1])
        if nu0 is None:
            if warm_start and self.nu is not None and np.all(np.isfinite(self.nu)) and np.all(np.asarray(self.nu) > 0.0):
                nu0 = self.nu                     # (a previous solve that ended on collapsed / non-finite prices is no warm start: cold instead)
            elif host and not np.all(u.c > 0):
                # start prices are a guess propagated through THIS rank's pools: rank 0's guess is everyone's
                nu0 = host.broadcast(start_prices(self.net, u) if rank == 0 else np.zeros(self.n), src=0)
            else:
                nu0 = start_prices(self.net, u)
        # start prices: derived from the (verified) utility, broadcast from rank 0, or the previous solution of an identical
        # solve -- identical on every rank by construction; prices the CALLER supplies are verified every time
        if host and nu0_given is not None:
            host.assert_identical(nu0, "the start prices")
        self._theta = {}
        self._trade_cache = None
        self._tol = tol
        kw = dict(max_evals=max_evals, memory=memory, iters_per_graph=iters_per_graph)
        total = dict(evals=0, iters=0, wall_seconds=0.0, device_seconds=0.0, rounds=0)
        general = _is_general(u)      # entries of the utility table: generic first-order iteration or the second-order path; no ties
        # (round 5 refused such a utility over ANY network with constant-sum pools, the K-asset table's included, for every method: what
        #  cannot be had is the tie loop -- a plain first-order leg and the second-order path, which smooths these pools, need no ties)
        # (the K-asset table's pools are in the second-order path too: the stableswap entry with its exact Hessian block, the constant-sum
        #  entry smoothed in price space -- csrc/phik.hpp: table_newton_kernel, gk_sum_newton_kernel)
        can_second = getattr(ctx, "second_order", False)
        if method not in _lib.METHODS:
            raise ValueError(f"method {method!r}: expected one of {sorted(_lib.METHODS)}")
        # auto: many stableswap pools -> second order straight away (first order needs thousands of evaluations there);
        # otherwise first order (with the active-set loop over constant-sum kinks), second order if that fails
        many_stable = n_stable >= AUTO_NEWTON_MIN_STABLE
        second_order = method == "newton" or (method == "auto" and can_second and many_stable)
        def first_order_leg(nu_from):
            """the first-order iteration from `nu_from` (with the active-set loop where the network has constant-sum pools), finished"""
            if n_sum == 0 or general:
                st = self._run(ctx, nu_from, total, tol=tol, method=_lib.METHODS["lbfgs"], **kw)
                nu, psi = self._solution_of(ctx, st, nu_from)
            else:
                self._kinks_settled = False
                st, nu, psi = self._solve_kinks(ctx, nu_from, tol, dict(kw, method=_lib.METHODS["lbfgs"]), kink_tol, max_rounds, total)
                if st["status"] == 1 and not self._kinks_settled:
                    st = dict(st, status=2)       # (the last leg converged on a REDUCED dual whose ties did not yield their fills: not a solution)
            self._finish(st, nu, psi, total)
            return st

        retry_near = False
        if not second_order:
            st = first_order_leg(nu0)
            # what decides is the CERTIFICATES of the point the run ended on (recomputed here, the worthless-component repair included), not
            # the device's verdict: a run that "stalled" at 1e-44 prices on a value-0 instance is optimal, and a run that reported
            # convergence while a whole region's prices collapsed (the device's test is value-weighted) is not (tools/fuzz_small.py,
            # fuzz_mid.py).  Uncertified: the second-order path from the start prices -- unless the device had converged and the
            # miss is a near one ("inaccurate", with its figures)
            near = st["status"] == 1 and max(self.gap, self.infeas) <= 100.0 * max(tol, 1e-12)
            # (a near miss on a SMALL network goes on all the same: a second-order solve of <= 512 tokens costs a millisecond or two, and
            #  "inaccurate at 1.02 x the tolerance" is what the round-6 table campaign's one failure in 360 was -- seed 703; should the
            #  second-order path not certify it either, the first-order point is solved for again below)
            retry_near = near and self.n <= 512
            if not (method == "auto" and can_second and self.status not in ("optimal", "infeasible") and (not near or retry_near)):
                return self.value
            second_order = True
            if self._dev_ties:
                self._clear_ties(ctx)
            self._theta = {}; self._trade_cache = None
        # warm start after a second-order solve on this context: nu0 = NULL tells the library to continue from its
        # own previous solution -- the prices AND (a multiple of) the final barrier weight, which is what turns the
        # ~18 steps of a cold solve into 4-7 (the parametric sweep of two-asset.py:34-100)
        cont = warm_start and nu0_given is None and self.nu is not None and (self.stats or {}).get("method") == _lib.METHODS["newton"]
        st = self._run(ctx, None if cont else nu0, total, tol=tol, method=_lib.METHODS["newton"], **kw)
        nu, psi = self._solution_of(ctx, st, nu0)
        self._finish(st, nu, psi, total)
        if method == "auto" and self.status not in ("optimal", "infeasible") and psi is not None and not retry_near and st.get("newton_steps", 0) > 0:
            # the second-order path ended without its certificates: ONE continuation from where it stopped -- the prices and a multiple of the
            # barrier weight it reached (cfmm_solve with nu0 = NULL).  Its end game on a partially filled constant-sum pool moves prices
            # below fp64 resolution, and whether the last steps centre or stall is decided by summation noise (tools/fuzz_table.py seeds
            # 1389, 1401: the same instance ends optimal in one run and stalled in the next); re-centring at a larger weight settles it.
            st2 = self._run(ctx, None, total, tol=tol, method=_lib.METHODS["newton"], **kw)
            nu2, psi2 = self._solution_of(ctx, st2, nu)
            if psi2 is not None:
                self._finish(st2, nu2, psi2, total)
        if method == "auto" and retry_near and self.status != "optimal":
            if self._dev_ties:
                self._clear_ties(ctx)
            self._theta = {}; self._trade_cache = None
            first_order_leg(nu0)                  # (the near miss stands: its point again, so that the device holds what this object reports)
            return self.value
        if method == "newton" and self.status not in ("optimal", "infeasible") and psi is not None and not general:
            # The barrier path asked for BY NAME and ended without its certificates (round 6; tools/fuzz_small.py seeds 1387, 1501): optima
            # where nothing trades, or a partially filled constant-sum pool decides, are what the first-order leg with its active-set
            # loop is for -- `auto` sends such problems there first; an explicit "newton" now hands it the prices the path ended on.
            # Kept only if that leg ends certified or at least closer: otherwise the path's own point is solved for again (the device's
            # buffers must hold the point this object reports).
            worst = lambda: max(abs(self.gap) / tol, self.infeas / tol) if np.isfinite(self.gap) and np.isfinite(self.infeas) else np.inf
            w_path, newton_steps = worst(), (self.stats or {}).get("newton_steps", 0)
            first_order_leg(np.asarray(nu, dtype=np.float64).copy())
            if self.status == "optimal" or worst() < w_path:
                self.stats["newton_steps"] = newton_steps
            else:
                if self._dev_ties:
                    self._clear_ties(ctx)
                self._theta = {}; self._trade_cache = None
                st = self._run(ctx, nu0, total, tol=tol, method=_lib.METHODS["newton"], **kw)
                nu, psi = self._solution_of(ctx, st, nu0)
                self._finish(st, nu, psi, total)
        return self.value

    @staticmethod
    def _solution_of(ctx, st, nu_start):
        """(nu, psi) of the run `st` -- or, for a run that ended in the library's numeric error, the start prices and NO net trade: the
        device's buffers then still hold the PREVIOUS solve's point, which must not be certified in this one's name"""
        if "numeric_error" in st:
            return np.asarray(nu_start, dtype=np.float64).copy(), None
        return ctx.get_solution()

    @staticmethod
    def _run(ctx, nu, total, **kw):
        try:
            st = ctx.solve(nu, **kw)
        except CfmmError as e:
            # an iteration that ran into non-finite numbers (a degenerate instance: prices hundreds of orders of magnitude apart) is a
            # solve that did not converge, not a crash of the caller: reported as "stalled", and `method="auto"` goes on to the
            # second-order path (tools/fuzz_table.py)
            # (ONLY the library's CFMM_E_NUMERIC -- "dual value is not finite", "non-finite value in the second-order iteration": an argument
            #  check that happens to mention "finite" -- a start price that is not positive and finite -- is the caller's error and is raised.
            #  the substring test swallowed those too, and _finish then certified the PREVIOUS solve's point)
            numeric = getattr(e, "code", None) == _lib.E_NUMERIC or (getattr(e, "code", None) is None and
                                                                     ("dual value is not finite" in str(e) or "non-finite value in the second-order iteration" in str(e)))
            if not numeric:
                raise
            st = dict(evals=0, iters=0, status=2, n_ranks=1, dual_value=float("nan"), primal_value=float("nan"), gap=float("inf"), infeas=float("inf"),
                      wall_seconds=0.0, device_seconds=0.0, pg=float("nan"), pool_subproblems=0, barrier_mu=0.0, newton_steps=0,
                      method=kw.get("method", 0), numeric_error=str(e))
        total["evals"] += st["evals"]; total["iters"] += st["iters"]; total["rounds"] += 1
        total["wall_seconds"] += st["wall_seconds"]; total["device_seconds"] += st["device_seconds"]
        return st

    def _clear_ties(self, ctx):
        ctx.set_ties(None, None)
        if "sum2" in self.net:
            ctx.set_pool_flags(POOL_SUM2, None)
        for (kd, k) in self.net.get("gk", {}):
            if kd == "sum" and hasattr(ctx, "set_pool_flagsG"):
                ctx.set_pool_flagsG(k, None)
        self._dev_ties = False

    def _kink_candidates(self, nu, kink_tol, banned, tied, loose=False):
        """constant-sum pools whose price ratio sits on one of their kinks: nearest kink, within kink_tol and well inside its half
        (`loose`: anywhere within kink_tol of it -- a guess for a leg that would not converge; see _solve_kinks).
        Two-asset pools: log nu_a - log nu_b = +-log gamma.  Pools of the K-asset table (arbitrage.py:73-74 over k > 2 tokens): with `lo`
        the pool's cheapest token, leg j is partially drained where gamma nu_j = nu_lo -- the same record with a = lo, b = j, tender a.
        Returns {(rank, k, pool, leg): record} (k = 2, leg = 0: the two-asset bucket): the pool's data travels with its key, so that
        every rank of a pool-sharded solve can build the same ties and the same fill recovery from the union of all ranks' candidates.
        A K-asset record carries the pool's WHOLE token list and reserves (`pidx`, `pR`): whether its kink still stands, what a switch
        record pays and the drain -> switch conversion are then functions of the record alone -- on every rank the same, whoever owns
        the pool (round 5 looked them up in the local bucket under the record's pool index: another rank's pool, or none -- ADVICE r5)."""
        rank = self._host.rank if self._host else 0
        out = {}
        if "sum2" in self.net:
            b = self.net["sum2"]
            r = np.log(nu[b["ia"]]) - np.log(nu[b["ib"]])
            lg = np.log(b["fee"])
            sgn = np.where(r < 0, 1, -1)                 # a->b kink at r = lg < 0, b->a kink at r = -lg > 0
            dist = np.abs(r - sgn * lg)
            near = (dist < kink_tol) & ((dist < 0.5 * np.abs(lg)) | (lg == 0.0) | loose)
            for i in np.flatnonzero(near):               # (only the pools on a kink: the scan itself is vectorised)
                i = int(i)
                key = (rank, 2, i, 0)
                if key not in tied and (key, int(sgn[i])) not in banned:
                    out[key] = dict(sgn=int(sgn[i]), ia=int(b["ia"][i]), ib=int(b["ib"][i]), fee=float(b["fee"][i]),
                                    Ra=float(b["Ra"][i]), Rb=float(b["Rb"][i]), loose=bool(loose))
        for (kd, k), b in self.net.get("gk", {}).items():
            if kd != "sum":
                continue
            lnu = np.log(nu[b["idx"]])                   # [k][m]
            lo = np.argmin(lnu, axis=0)
            m = lnu.shape[1]
            lg = np.log(b["fee"])
            r = lnu[lo, np.arange(m)][None, :] - lnu     # log nu_lo - log nu_j  (<= 0)
            dist = np.abs(r - lg[None, :])
            near = (dist < kink_tol) & ((dist < 0.5 * np.abs(lg)[None, :]) | (lg == 0.0)[None, :] | loose)
            near[lo, np.arange(m)] = False
            for j, i in zip(*np.nonzero(near)):
                i, j = int(i), int(j)
                key = (rank, k, i, j)
                if key not in tied and (key, 1) not in banned:
                    out[key] = dict(sgn=1, ia=int(b["idx"][lo[i], i]), ib=int(b["idx"][j, i]), fee=float(b["fee"][i]),
                                    Ra=float(b["R"][lo[i], i]), Rb=float(b["R"][j, i]), loose=bool(loose), leg_lo=int(lo[i]),
                                    pidx=[int(x) for x in b["idx"][:, i]], pR=[float(x) for x in b["R"][:, i]])
            # the pool's OTHER kind of kink: two tokens tied for cheapest (nu_a = nu_b) -- which of them pays for the drained tokens
            # switches there, and the whole payment with it; at the optimum it is split.  Record: a = the token the device makes pay (the
            # first of the equals), b = the other; "sgn" 0 marks it; the amount P / gamma is filled in when the fills are recovered
            srt = np.argsort(lnu, axis=0, kind="stable")
            a1, a2 = srt[0], srt[1]
            gap = lnu[a2, np.arange(m)] - lnu[a1, np.arange(m)]
            # (only where the pool PAYS something at these prices -- some token worth more than the cheapest after the fee: a pool that
            #  does not trade has no kink there, and tying its two cheapest prices only takes a degree of freedom from the dual)
            pays = (lnu + lg[None, :] > lnu[a1, np.arange(m)][None, :] + 1e-12).sum(axis=0) > 0
            sw = pays & (gap < kink_tol) & ((gap < 0.5 * np.abs(lg)) | (lg == 0.0) | loose)
            for i in np.flatnonzero(sw):
                i = int(i)
                ja, jb = sorted((int(a1[i]), int(a2[i])))         # (equal prices: the device takes the lower leg as the payer)
                key = (rank, k, i, 100 + 10 * ja + jb)
                if key not in tied and (key, 0) not in banned:
                    out[key] = dict(sgn=0, ia=int(b["idx"][ja, i]), ib=int(b["idx"][jb, i]), fee=float(b["fee"][i]), Ra=0.0, Rb=0.0,
                                    loose=bool(loose), leg_a=ja, leg_b=jb, pool=i, k=k,
                                    pidx=[int(x) for x in b["idx"][:, i]], pR=[float(x) for x in b["R"][:, i]])
        if self._host:                                   # the union over ranks, identical everywhere
            merged = {}
            for part in self._host.allgather(out):
                merged.update(part)
            out = merged
        return dict(sorted(out.items()))

    def _solve_kinks(self, ctx, nu, tol, kw, kink_tol, max_rounds, total):
        """A constant-sum pool whose optimum is a partial fill sits on a kink of the dual (all three
        shipped scripts do this).  Active-set loop: run the device solver in short legs; tie the two
        prices of every pool found on a kink (a linear equality in log-price) and skip it in the
        kernels; once the (now smooth) reduced dual has converged recover the fill fractions;
        release ties whose fill leaves (0,1).  Pool-sharded: every decision below is a function of the
        all-reduced prices / psi and of the all-gathered candidates, hence identical on every rank.
        Round 5: the K-asset table's constant-sum pools take part leg by leg (a tied LEG is left out of the evaluation)."""
        rank = self._host.rank if self._host else 0
        m2 = len(self.net["sum2"]["Ra"]) if "sum2" in self.net else 0
        gks = {k: b["R"].shape for (kd, k), b in self.net.get("gk", {}).items() if kd == "sum"}
        tied, banned = {}, set()
        # legs: short on small networks (a solve stuck on a kink gains nothing from more evaluations: the reference's own
        # instances spent 300 of their 315 evaluations that way), longer where an evaluation covers many pools
        budget = min(kw["max_evals"], 40 if self.m <= 1000 else 100)
        st = None
        psi = None
        for _ in range(8 * max_rounds):
            if total["evals"] >= 4 * kw["max_evals"]:
                break
            ties = _Ties(self.n)
            flags = np.zeros(m2, dtype=np.int32)
            flagsg = {k: np.zeros(shp, dtype=np.int32) for k, shp in gks.items()}
            for key, rec in list(tied.items()):
                if ties.tie(rec["ia"], rec["ib"], rec["sgn"] * np.log(rec["fee"])):
                    if key[0] == rank:
                        if key[1] == 2:
                            flags[key[2]] = 1
                        elif key[3] < 100 or key[3] >= 200:           # (a switch record ties two prices and flags nothing)
                            flagsg[key[1]][_drain_leg(key[3]), key[2]] = 1
                else:
                    del tied[key]; banned.add((key, rec["sgn"]))
            self._dev_ties = True
            if tied:
                grp, off, ng = ties.groups()
                ctx.set_ties(grp, off)
                if m2:
                    ctx.set_pool_flags(POOL_SUM2, flags)
                for k, f in flagsg.items():
                    ctx.set_pool_flagsG(k, f)
                st = self._run(ctx, nu, total, tol=0.01 * tol, pg_rule=1, **dict(kw, max_evals=budget))
            else:
                ctx.set_ties(None, None)
                if m2:
                    ctx.set_pool_flags(POOL_SUM2, None)
                for k in flagsg:
                    ctx.set_pool_flagsG(k, None)
                st = self._run(ctx, nu, total, tol=tol, **dict(kw, max_evals=budget))
            if "numeric_error" in st:
                return st, nu, None        # (the leg ran into non-finite numbers: its start prices, no net trade -- _finish reports "stalled")
            nu, psi = ctx.get_solution()
            if not (np.all(np.isfinite(nu)) and np.all(nu > 0.0)):
                break                      # a price has collapsed (a token to sell that no pool lists: the dual is unbounded) -- _finish says "infeasible"
            if st["status"] == 1:
                if not tied:
                    self._kinks_settled = True
                    return st, nu, psi
                # a K-asset record is a statement about the pool's CHEAPEST token (leg j drained where gamma nu_j = nu_lo; the two
                # cheapest tied): if the prices have since made another token the cheapest, the flagged leg is no longer on a kink --
                # it would be drained outright through the new cheapest token, value the reduced dual does not see (tools/fuzz_table.py:
                # a "certified" optimum below SLSQP's feasible point).  Such ties are dropped and the leg repeated
                stale = [k for k, rec in tied.items() if k[1] != 2 and not self._kink_still_stands(nu, k, rec)]
                if stale:
                    for k in stale:
                        del tied[k]
                    continue
                tied = self._canonical_switches(tied, banned)
                n_before = len(tied)
                tied = self._split_payers(tied, banned)
                if len(tied) != n_before:     # (new records tie new prices: a leg with them first)
                    continue
                self._refresh_switches(nu, tied)
                theta, ok = self._recover_fills(nu, psi, tied, tol)
                if os.environ.get("CFMM_KINK_TRACE"):
                    print("[kinks] status", st["status"], "ok", ok, "records", [(k[1:], r["sgn"], r["ia"], r["ib"], round(float(r.get("Rb", 0.0)), 4), round(float(theta[k]), 6)) for k, r in tied.items()], file=sys.stderr)
                bad = [k for k in tied if not (1e-9 < theta[k] < 1 - 1e-9) or (tied[k]["sgn"] == 0 and not tied[k]["Rb"] > 0.0)]      # (a switch record with nothing to move)
                # Switch records of ONE pool that move its payment away from the SAME leg share that payment: their fills live on a
                # simplex, sum theta < 1, not in a box.  Three tokens tied for cheapest (two records from one leg) came back from the
                # box-bounded least squares with fills 0.40 + 0.91: the leg "paid" -30 % of the payment, i.e. received a token it tenders,
                # fee-free -- every token balanced, both certificates met, and a value 1.3e-3 ABOVE the optimum (round 6, the dual
                # referee on tools/fuzz_table.py seeds 1059, 1358: the first FALSELY CERTIFIED points it found).  Such a group means
                # its source leg is not a payer at all: its records are released like any fill outside (0, 1).
                groups = {}
                for k in tied:
                    if tied[k]["sgn"] == 0 and k not in bad:
                        groups.setdefault((k[0], k[1], k[2], tied[k]["ia"]), []).append(k)
                for k in tied:            # ... and so do the records of ONE leg paid for by several of the pool's cheapest tokens
                    if tied[k]["sgn"] == 1 and k[1] != 2 and k not in bad:
                        groups.setdefault((k[0], k[1], k[2], "leg", _drain_leg(k[3])), []).append(k)
                for ks in groups.values():
                    if len(ks) > 1 and sum(theta[k] for k in ks) >= 1 - 1e-9:
                        bad.extend(ks)
                if ok and not bad:
                    self._theta = {k: (tied[k], theta[k]) for k in tied}
                    self._kinks_settled = True
                    return st, nu, psi
                if bad:                   # fully on / fully off after all: back to bang-bang
                    for k in bad:
                        rec = tied.pop(k)
                        banned.add((k, rec["sgn"]))
                        # a guessed kink that carries no trade: the pool's fee band is narrower than the leg could
                        # resolve and the optimum sits on its OTHER kink (arbitrage.py / liquidation.py: fee 0.999)
                        if k[1] == 2 and rec.get("loose") and theta[k] <= 1e-9 and (k, -rec["sgn"]) not in banned:
                            tied[k] = dict(rec, sgn=-rec["sgn"], loose=False)
                        # the K-asset analogue: a guessed DRAIN kink (gamma nu_j = nu_lo) that carries no trade -- inside the fee band the
                        # pool's other kink on that pair is the SWITCH (nu_j = nu_lo: j pays alongside lo), tools/fuzz_table.py seed 256
                        elif k[1] != 2 and rec["sgn"] == 1 and k[3] < 100 and theta[k] <= 1e-9:
                            ja, jb = sorted((rec["leg_lo"], k[3]))
                            k2 = (k[0], k[1], k[2], 100 + 10 * ja + jb)
                            if k2 not in tied and (k2, 0) not in banned:
                                tied[k2] = dict(sgn=0, ia=int(rec["pidx"][ja]), ib=int(rec["pidx"][jb]), fee=rec["fee"], Ra=0.0, Rb=0.0,
                                                loose=False, leg_a=ja, leg_b=jb, pool=k[2], k=k[1], pidx=rec["pidx"], pR=rec["pR"])
                    tied = dict(sorted(tied.items()))
                    continue
            new = self._kink_candidates(nu, kink_tol, banned, tied)
            # nothing within the tolerance although the leg did not converge: look further out before spending another leg
            # (a wrongly tied pool shows as a fill outside (0,1) and is released again)
            wide = kink_tol
            while not new and st["status"] != 1 and wide < 0.05:
                wide *= 10
                new = self._kink_candidates(nu, wide, banned, tied, loose=True)
            if new:
                tied.update(new)
                tied = dict(sorted(tied.items()))
                banned = {bn for bn in banned if bn[0] in new}     # bans expire when the active set changes
            elif st["status"] == 3 and budget < kw["max_evals"]:
                budget = min(kw["max_evals"], 2 * budget)           # no kink in sight: longer legs
            elif kink_tol < 0.05:
                kink_tol *= 10
            else:
                break
        return st, nu, psi

    def _kink_still_stands(self, nu, key, rec):
        lnu = np.log(nu[np.asarray(rec["pidx"], dtype=np.int64)])        # (the pool's own token list: the record's, not a local bucket's)
        lo = float(lnu.min())
        legs = (rec["leg_a"], rec["leg_b"]) if rec["sgn"] == 0 else (rec["leg_lo"],)
        return all(lnu[j] <= lo + 1e-9 for j in legs)

    @staticmethod
    def _canonical_switches(tied, banned):
        """A K-asset constant-sum pool with SEVERAL tokens tied for cheapest: the switch records that tie them were found pair by pair, in
        rounds whose cheapest token differed -- (0, 3) in one, (2, 3) in another.  The device makes ONE leg pay, the lowest of the tied
        legs (csrc/phik.hpp: "ties: the first"), and a record moves a share of THAT leg's payment: a record rooted at another leg moves
        a payment that leg never made (round 6, tools/fuzz_table.py seed 1358: leg 2 "paid" -57 %, every token balanced, both
        certificates met, the value 1.1e-3 above the optimum -- caught by the dual referee).  Canonical form: for the tied legs T of a
        pool, one record (min T -> t) per t in T; same union of tied tokens, so the price ties on the device are what they were."""
        pools = {}
        for k, rec in tied.items():
            if rec["sgn"] == 0:
                pools.setdefault((k[0], k[1], k[2]), []).append(k)
        out = dict(tied)
        for (r, kk, i), ks in pools.items():
            legs = sorted({tied[k]["leg_a"] for k in ks} | {tied[k]["leg_b"] for k in ks})
            want = {(r, kk, i, 100 + 10 * legs[0] + t): t for t in legs[1:]}
            if set(want) == set(ks):
                continue
            proto = tied[ks[0]]
            for k in ks:
                del out[k]
            for k2, t in want.items():
                if (k2, 0) in banned:
                    continue
                out[k2] = dict(proto, ia=int(proto["pidx"][legs[0]]), ib=int(proto["pidx"][t]), leg_a=legs[0], leg_b=t, Ra=0.0, Rb=0.0, loose=False)
        return dict(sorted(out.items()))

    @staticmethod
    def _split_payers(tied, banned):
        """A K-asset constant-sum pool on BOTH kinds of kink at once: leg j partly drained (gamma nu_j = nu_lo) while several tokens are tied
        for cheapest.  The drain record says who pays for the fill -- ONE token, the cheapest when the kink was found -- but at the optimum
        the payment may be split between the tied tokens, and the switch record cannot do it: what it moves is the payment for the legs
        drained OUTRIGHT, which is zero here (round 6, tools/fuzz_table.py seed 2595: the loop cycled through four tie sets, each
        unbalanced, for 48 rounds).  So the leg gets one record per cheapest token that could pay for it: the same fill vector with
        another payer, key 200 + 10 t + j; their fills share the leg, sum theta < 1 (the simplex test below)."""
        pools = {}
        for k, rec in tied.items():
            if rec["sgn"] == 0:
                pools.setdefault((k[0], k[1], k[2]), set()).update((rec["leg_a"], rec["leg_b"]))
        if not pools:
            return tied
        out = dict(tied)
        for k, rec in tied.items():
            if rec["sgn"] != 1 or k[1] == 2 or (k[0], k[1], k[2]) not in pools:
                continue
            T, j = pools[(k[0], k[1], k[2])], _drain_leg(k[3])
            if rec["leg_lo"] not in T:
                continue
            for t in sorted(T):
                k2 = (k[0], k[1], k[2], 200 + 10 * t + j)
                if t in (rec["leg_lo"], j) or k2 in out or (k2, 1) in banned or (k[0], k[1], k[2], j) in out and out[(k[0], k[1], k[2], j)]["leg_lo"] == t:
                    continue
                out[k2] = dict(rec, ia=int(rec["pidx"][t]), Ra=float(rec["pR"][t]), leg_lo=t, loose=False)
        return dict(sorted(out.items()))

    def _refresh_switches(self, nu, tied):
        """the payment a K-asset constant-sum pool's cheapest token makes at the prices nu (what the device evaluated): the reserves of
        every token drained there over the fee, tied legs' fills aside -- the amount a `switch` record moves to the other cheapest token"""
        for key, rec in tied.items():
            if rec["sgn"] != 0:
                continue
            i = rec["pool"]
            p = nu[np.asarray(rec["pidx"], dtype=np.int64)]
            R = np.asarray(rec["pR"], dtype=np.float64)
            lo = rec["leg_a"]
            drained = rec["fee"] * p > p[lo]
            drained[lo] = False
            for k2, r2 in tied.items():                           # legs of this pool tied on a drain kink: left out by the device
                if r2["sgn"] == 1 and k2[0] == key[0] and k2[1] == rec["k"] and k2[2] == i:
                    drained[_drain_leg(k2[3])] = False
            rec["Rb"] = float(R[drained].sum() / rec["fee"])

    def _fill_vector(self, rec):
        """net trade of the tied constant-sum pool `rec` at full fill in its kink's direction"""
        d = np.zeros(self.n)
        a, bb, g = rec["ia"], rec["ib"], rec["fee"]
        if rec["sgn"] == 0:   # a K-asset pool whose two cheapest tokens are tied: the payment P / gamma moves from a to b
            d[a] = rec["Rb"]; d[bb] = -rec["Rb"]                  # (Rb holds P / gamma: _refresh_switches)
            return d
        if rec["sgn"] > 0:    # tender a, drain b
            d[a] = -rec["Rb"] / g; d[bb] = rec["Rb"]
        else:                 # tender b, drain a
            d[bb] = -rec["Ra"] / g; d[a] = rec["Ra"]
        return d

    def _recover_fills(self, nu, psi, tied, tol):
        """theta in [0,1]^K with  (psi + h + sum_k theta_k d_k)_j = 0 on every token that must balance
        (EQ tokens, GE tokens priced above their bound) and >= 0 on GE tokens at their bound."""
        u = self.utility
        keys = list(tied)
        D = np.stack([self._fill_vector(tied[k]) for k in keys], axis=1)       # n x K
        r = psi + u.h
        at_bound = (u.ctype == GE) & (nu <= u.c * (1 + 1e-9))
        must = (u.ctype == EQ) | ((u.ctype == GE) & ~at_bound)
        touched = np.abs(D).sum(axis=1) > 0
        rows = must & touched
        A = D[rows] * nu[rows, None]
        rhs = -r[rows] * nu[rows]
        if A.shape[0] == 0:
            th = np.full(len(keys), 0.5)
        elif len(keys) == 1:              # one tied pool (every shipped script): the bounded least squares in closed form
            a1 = A[:, 0]
            th = np.array([min(1.0, max(0.0, float(a1 @ rhs) / max(float(a1 @ a1), 1e-300)))])
        else:
            th = np.linalg.lstsq(A, rhs, rcond=None)[0]
            if th.min() < 0.0 or th.max() > 1.0:          # a bound is active: the bounded problem proper
                try:
                    from scipy.optimize import lsq_linear
                    th = lsq_linear(A, rhs, bounds=(0.0, 1.0), tol=1e-14).x
                except Exception:
                    th = np.clip(th, 0.0, 1.0)
        tot = r + D @ th
        scale = max(1.0, float(np.abs(nu * (np.abs(psi) + np.abs(u.h))).sum()))
        res_eq = np.abs(nu[must] * tot[must]).sum() / scale if must.any() else 0.0
        ge_b = at_bound
        res_ge = np.maximum(-(nu[ge_b] * tot[ge_b]), 0.0).sum() / scale if ge_b.any() else 0.0
        ok = (res_eq <= 10 * tol) and (res_ge <= 10 * tol)
        return dict(zip(keys, th)), ok

    def _listed_tokens(self):
        """per token: some pool (of any rank) lists it"""
        ls = getattr(self, "_listed", None)
        if ls is None:
            ls = np.zeros(self.n, dtype=bool)
            for _, idx in self._pool_token_arrays():
                ls[idx.ravel()] = True
            if self._host:
                ls = np.any(np.stack(self._host.allgather(ls)), axis=0)
            self._listed = ls
        return ls

    def _max_reserve(self):
        """largest reserve of the network (all ranks of a pool-sharded problem: the floor of a relative figure must not differ between them)"""
        mr = getattr(self, "_rmax", None)
        if mr is None:
            mr = 0.0
            for key in KIND2:
                if key in self.net and len(self.net[key]["Ra"]):
                    mr = max(mr, float(self.net[key]["Ra"].max()), float(self.net[key]["Rb"].max()))
            for b in list(self.net.get("gn", {}).values()) + list(self.net.get("gk", {}).values()):
                if b["R"].size:
                    mr = max(mr, float(b["R"].max()))
            if self._host:
                mr = float(max(self._host.allgather(mr)))
            self._rmax = mr
        return mr

    def _pool_token_arrays(self):
        """(key, idx [k][m]) of every non-empty bucket: the token ids of its pools, leg by leg"""
        out = []
        for key in KIND2:
            if key in self.net and len(self.net[key]["Ra"]):
                out.append((key, np.stack([self.net[key]["ia"], self.net[key]["ib"]])))
        for k, b in self.net.get("gn", {}).items():
            if b["R"].shape[1]:
                out.append((k, b["idx"]))
        for key, b in self.net.get("gk", {}).items():
            if b["R"].shape[1]:
                out.append((key, b["idx"]))
        return out

    def _recover_worthless(self, st, nu, psi, total):
        """Tokens worth NOTHING at the optimum -- no chain of pools leads from them to anything the utility values: a disconnected
        component, a target no pool lists -- have prices that run to zero TOGETHER: the dual is flat along that ray, the ratios between
        them never settle, and the pools among them trade at whatever ratio the last iterate had (found by tools/fuzz_small.py: such
        instances came back "infeasible" although the program is feasible -- cvxpy solves them).  The primal side of that degeneracy
        is simple: a pool ALL of whose tokens are worthless can be left untouched at no cost in the objective, and a worthless token that
        must leave the trader's hands (an equality, liquidation.py:77-80) can be given to any pool that lists it (Delta > 0 with
        Lambda = 0 only raises the pool's trading function: arbitrage.py:60,63-74).  Done here, on the host, on the tenders read back;
        accepted only if the certificates then hold.  Returns True if they do (and _finish has been re-run on the repaired point)."""
        u = self.utility
        if _is_general(u):
            return False
        # "worthless": a price far below the largest.  How far is not knowable in advance (the iteration stops somewhere down the
        # ray) -- and need not be: the repair is accepted only if BOTH certificates hold afterwards, and leaving out a pool whose
        # arbitrage value is not negligible breaks the gap.  So: the strictest threshold first, looser ones after it
        for thr in (1e-9, 1e-7, 1e-5):
            W = nu <= thr * float(nu.max())
            if W.any() and self._recover_worthless_at(W, st, nu, psi, total):
                return True
        return False

    def _recover_worthless_at(self, W, st, nu, psi, total):
        u = self.utility
        tr = self._trades()                               # (the tied pools' fills included)
        # psi of the repaired point is re-summed from the tenders that are handed out, not patched: down a flat ray the prices end
        # tens of orders of magnitude apart, and there the evaluation's psi and the tender kernel's need not agree on a worthless pool
        psi2 = np.zeros(self.n)
        psi_dropped = np.zeros(self.n)                    # net trade of the pools left untouched: their arbitrage value stays in the DUAL
        zeroed, lister = {}, {}
        arrays = self._pool_token_arrays()
        for key, idx in arrays:
            allw = W[idx].all(axis=0)
            for leg in range(idx.shape[0]):               # a pool to give token j to: the first that lists it
                for pos in np.flatnonzero(W[idx[leg]]):
                    lister.setdefault(int(idx[leg, pos]), (key, leg, int(pos)))
            d, l = tr[key]
            keep = ~allw
            np.add.at(psi2, idx[:, keep].ravel(), (l - d)[:, keep].ravel())
            if allw.any():
                zeroed[key] = allw
                np.add.at(psi_dropped, idx[:, allw].ravel(), (l - d)[:, allw].ravel())
        r = psi2 + u.h
        give = np.where(W & (u.ctype == EQ) & (r > 0.0), r, 0.0)
        for j in np.flatnonzero(give):
            if int(j) not in lister:
                return False                              # (no pool lists it: truly infeasible)
        if not zeroed and not give.any():
            return False
        # the dual value at nu counts EVERY pool's arbitrage value, the untouched ones' included: the repaired point's gap is taken against
        # that, not against a dual rebuilt from the repaired psi (with the zeroed pools' nu'(L - D) missing from both sides a
        # repair at the loosest threshold could pass the gap test it should fail)
        self._repair_dual = float((nu - u.c) @ u.h + nu @ (psi2 + psi_dropped))
        psi2 = psi2 - give
        saved = (self._theta, self._trade_cache)
        names = ("value", "dual_value", "gap", "infeas", "nu", "psi", "status", "stats")
        before = {k: getattr(self, k, None) for k in names}
        self._theta = {}                                  # (psi2 already holds the fills)
        self._finish(st, nu, psi2, total, _recovering=True)
        self._repair_dual = None
        if self.status != "optimal":                      # not a repair: everything back as the solve left it
            self._theta, self._trade_cache = saved
            for k, v in before.items():
                setattr(self, k, v)
            return False
        # the tenders of the repaired point: the untouched pools at zero, the gifts on top
        tr = {key: (d.copy(), l.copy()) for key, (d, l) in tr.items()}
        for key, allw in zeroed.items():
            tr[key][0][:, allw] = 0.0; tr[key][1][:, allw] = 0.0
        for j in np.flatnonzero(give):
            key, leg, pos = lister[int(j)]
            tr[key][0][leg, pos] += give[j]
        self._theta = saved[0]
        self._trade_cache = tr
        self.stats["worthless_tokens"] = int(W.sum())
        return True

    def _finish(self, st, nu, psi, total, _recovering=False):
        u = self.utility
        if psi is None:                       # a run that ended in a numeric error: nothing to certify
            self.value = self.dual_value = float("nan")
            self.gap = self.infeas = float("inf")
            self.nu, self.psi = nu, np.full(self.n, np.nan)
            self.status = "stalled"
            self.stats = dict(st); self.stats.update(total)
            self.stats["pool_subproblems"] = total["evals"] * self.m
            return self
        if self._theta:
            psi = psi.copy()
            for rec, th in self._theta.values():
                psi += th * self._fill_vector(rec)
        if _is_general(u):
            # a utility with entries of the table: value, dual value and certificates are the device's (Fenchel-Young gap of
            # the pair (nu, psi): csrc/lbfgs_rules.hpp); recomputed here in NumPy only by the tests
            self.value = float(st["primal_value"]); self.dual_value = float(st["dual_value"])
            self.gap = abs(float(st["gap"])); self.infeas = float(st["infeas"])
            self.nu, self.psi = nu, psi
            self.status = _lib.STATUS.get(st["status"], f"error {st['status']}")
            tolx = max(self._tol, 1e-12) * (1 + 1e-6) + 1e-15
            if self.status == "optimal" and not (self.gap <= tolx and self.infeas <= tolx):
                self.status = "inaccurate"
            self.stats = dict(st)
            self.stats.update(total)
            return
        plain = getattr(u, "_plain", None)
        if plain is None:                    # h == 0 and psi >= 0 everywhere (arbitrage.py:57,77): a shorter certificate check
            plain = u._plain = bool(not u.h.any() and not u.ctype.any())
        r = psi if plain else psi + u.h
        self.value = float(u.c @ psi)
        nu_psi = float(nu @ psi) if plain else None           # (plain: (nu - c)'psi = nu'psi - c'psi, two dot products serve three quantities)
        cs = nu_psi - self.value if plain else float((nu - u.c) @ r)      # complementary slackness (nu - c)'(psi + h)
        if st.get("method") == _lib.METHODS["newton"]:
            # psi is the barrier-smoothed primal point (strictly inside every pool's trading set); the dual value
            # and the gap against it were computed on the device from an exact evaluation at nu
            self.dual_value = float(st["dual_value"])
            self.gap = abs(float(st["gap"]))
            if _recovering:                   # (a repaired primal point: the duality gap against the device's exact dual value at nu)
                self.gap = abs(self.dual_value - self.value) / max(1.0, abs(self.dual_value))
        else:
            # sum_i arb_i = nu'psi_pools; tied pools trade value-neutrally at their kink prices
            self.dual_value = nu_psi if plain else float((nu - u.c) @ u.h + nu @ psi)
            self.gap = abs(cs) / max(1.0, abs(self.dual_value))
            if _recovering and getattr(self, "_repair_dual", None) is not None:      # (a repaired primal point against the dual value of ALL pools)
                self.dual_value = self._repair_dual
                self.gap = abs(self.dual_value - self.value) / max(1.0, abs(self.dual_value))
        floor = 1e-12 * self._max_reserve()       # (trades of rounding size at a no-arbitrage optimum: noise over noise is not an infeasibility)
        if plain:
            lo, hi = float(psi.min()), float(psi.max())
            viol = max(-lo, 0.0)
            scale = max(hi, -lo, floor, 1e-300)
        else:
            viol = float(np.where(u.ctype == GE, np.maximum(-r, 0.0), np.where(u.ctype == EQ, np.abs(r), 0.0)).max())
            scale = max(float(np.abs(psi).max()), float(np.abs(u.h).max()), floor, 1e-300)
        self.infeas = viol / scale
        self.nu, self.psi = nu, psi
        self.status = _lib.STATUS.get(st["status"], f"error {st['status']}")
        # "optimal" means what the caller asked for: both certificates at the requested tolerance (the recomputation
        # above repeats the device's sums in another order: allow it rounding, not a looser bar)
        tolx = max(self._tol, 1e-12) * (1 + 1e-6) + 1e-15
        if self.gap <= tolx and self.infeas <= tolx:
            self.status = "optimal"
        elif not _recovering and self._host is None and self._recover_worthless(st, nu, psi, total):
            return self                       # (the certificates hold once the worthless component is taken out: _finish has run again)
        else:
            # the certificates do not hold.  A token that MUST leave the trader's hands (an equality with h > 0, liquidation.py:77-80)
            # and that no pool lists cannot: the program is infeasible, and is reported as that.  (Round 5: this used to be read off a
            # price collapsing to zero -- which is also what a WORTHLESS token's price does in a feasible program; any listed token can be
            # given to a pool, so the structural test is the whole of it.)  Otherwise the device's verdict stands, or "inaccurate" if it
            # believed it had converged
            stuck = (u.ctype == EQ) & (u.h > 0) & ~self._listed_tokens()
            if stuck.any():
                self.status = "infeasible"
            elif self.status == "optimal":
                self.status = "inaccurate"
        self.stats = dict(st)
        self.stats.update(total)
        self.stats["pool_subproblems"] = total["evals"] * self.m
        return self

    # -- result read-back (arbitrage.py:84, two-asset.py:94-100) --------------------------------
    def _trades(self):
        if self._trade_cache is None:
            ctx = self._ensure_ctx()
            tr = {}
            for key, kind in KIND2.items():
                if key in self.net:
                    tr[key] = ctx.get_trades2(kind, len(self.net[key]["Ra"]))
            for k, b in self.net.get("gn", {}).items():
                tr[k] = ctx.get_tradesN(k, b["R"].shape[1])
            for (kind, k), b in self.net.get("gk", {}).items():
                tr[(kind, k)] = ctx.get_tradesG(_lib.POOLK[kind], k, b["R"].shape[1])
            if self._theta:
                rank = self._host.rank if self._host else 0
                for (r, k, i, j), (rec, th) in self._theta.items():
                    if r != rank:                      # (another rank's pool: its tenders are read back there)
                        continue
                    full = self._fill_vector(rec)
                    if k == 2 and "sum2" in tr:
                        d, l = tr["sum2"]
                        y = th * np.array([full[rec["ia"]], full[rec["ib"]]])
                        d[:, i] = np.maximum(-y, 0.0); l[:, i] = np.maximum(y, 0.0)
                    elif ("sum", k) in tr and rec["sgn"] == 0:      # two cheapest tokens tied: theta of the payment moves from leg a to leg b
                        d, l = tr[("sum", k)]
                        d[rec["leg_a"], i] -= th * rec["Rb"]
                        d[rec["leg_b"], i] += th * rec["Rb"]
                    elif ("sum", k) in tr:             # a tied LEG of a K-asset pool: theta R_j received, paid for by the pool's cheapest token
                        d, l = tr[("sum", k)]
                        l[_drain_leg(j), i] += th * rec["Rb"]
                        d[rec["leg_lo"], i] += th * rec["Rb"] / rec["fee"]
            self._trade_cache = tr
        return self._trade_cache

    def bucket_trades(self, key):
        """(delta, lambda), slot-major [k][m], of one bucket ('cp2', 'w2', 'sum2', 'curve2', 'pow2' or a pool size)"""
        return self._trades()[key]

    def _per_pool(self, which):
        if self.where is None:
            raise CfmmError("per-pool lists need a Problem built from pool lists; use bucket_trades()")
        tr = self._trades()
        return [tr[key][which][:, pos].copy() for key, pos in self.where]

    @property
    def deltas(self):
        return self._per_pool(0)

    @property
    def lambdas(self):
        return self._per_pool(1)

    def close(self):
        for p in (getattr(self, "_batch_workers", None) or [])[1:]:
            p.close()
        self._batch_workers = None
        if self.ctx is not None:
            self.ctx.close(); self.ctx = None; self._uploaded = False
            self._dev_utility = None; self._dev_ties = False
