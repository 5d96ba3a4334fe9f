"""Synthetic CFMM-network generator; NumPy `default_rng(seed)`.

The reference ships only three literal toy instances (/root/reference/arbitrage.py:5-36,
liquidation.py:5-36, two-asset.py:7-32); BASELINE.json's configs 2-5 are synthetic
random-reserve networks, generated here so that bench.py, the tests and the CPU baseline
all see bit-identical inputs for a given (config, seed).

Recipe: latent token prices pi_j = exp(N(0,1)); per pool a distinct token tuple (uniform, or
Zipf hub-weighted), pool value L = exp(N(ln 1e3, 1.5^2)), reserves R_k = L*w_k/pi_k with a
2 % log-normal mispricing on the first leg; fee drawn from {0.997, 0.999, 0.9995, 0.99};
Balancer weights from small-denominator rationals; market values c_j = pi_j*exp(N(0,0.01^2)).
"""
import numpy as np

FEES = np.array([0.997, 0.999, 0.9995, 0.99])
W2_CHOICES = np.array([0.5, 0.8, 0.6, 0.2, 0.4])       # w_a of a 2-asset weighted pool


def _pairs(rng, n, m, zipf_s=None):
    """m ordered pairs (a, b), a != b."""
    if zipf_s is None:
        a = rng.integers(0, n, size=m)
        b = rng.integers(0, n - 1, size=m)
    else:
        pmf = 1.0 / np.arange(1, n + 1) ** zipf_s
        pmf /= pmf.sum()
        a = rng.choice(n, size=m, p=pmf)
        b = rng.choice(n - 1, size=m, p=None)
    b = b + (b >= a)
    return a.astype(np.int32), b.astype(np.int32)


def make_network(n_tokens, m_cp2=0, m_w2=0, m_gn=0, m_curve2=0, seed=0, zipf_s=None,
                 gn_sizes=(3, 8), mispricing=0.02, pool_seed=None, m_pow2=0, m_gk_stable=0, m_gk_sum=0, gk_sizes=(3, 4), peg=4):
    """Returns a dict of SoA buckets:

      prices   : latent pi[n]
      c        : market values c[n]                      (arbitrage utility, arbitrage.py:31-36)
      cp2      : dict(Ra, Rb, fee, ia, ib)               constant product   (arbitrage.py:68-70)
      w2       : dict(Ra, Rb, fee, wa, ia, ib)           2-asset weighted geo-mean (wb = 1-wa)
      gn       : dict(size -> dict(R[k,m], w[k,m], idx[k,m], fee[m]))  n-asset weighted geo-mean,
                 one slot-major ("size-class SoA") bucket per pool size  (arbitrage.py:65)
      curve2   : dict(Ra, Rb, fee, alpha, ia, ib)        phi = x + y - alpha/(xy)
      gk       : dict(("stable" | "sum", k) -> dict(idx[k,m], R[k,m], fee[m], param[m]))  the K-asset table's buckets
                 (csrc/phik.hpp): n-asset stableswap sum x - alpha / prod x among the tokens of one peg group, n-asset constant
                 sum over arbitrary tokens (distinct market prices: its LP vertex is then not degenerate)
    """
    rng = np.random.default_rng(seed)
    n = n_tokens
    pi = np.exp(rng.normal(0.0, 1.0, n))
    if m_curve2 or m_gk_stable:   # peg groups of 4 tokens (`peg`: of that many -- stableswap baskets of up to 8 tokens)
        pi = pi[(np.arange(n) // peg) * peg] * np.exp(rng.normal(0.0, 0.002, n))
    c = pi * np.exp(rng.normal(0.0, 0.01, n))
    out = dict(n_tokens=n, prices=pi, c=c, seed=seed)
    if pool_seed is not None:      # same tokens / prices / market values, a different draw of pools
        rng = np.random.default_rng([seed, 7919, int(pool_seed)])   # (one shard of a sharded network)

    def value(m):
        return np.exp(rng.normal(np.log(1e3), 1.5, m))

    if m_cp2:
        ia, ib = _pairs(rng, n, m_cp2, zipf_s)
        L = value(m_cp2)
        out["cp2"] = dict(
            Ra=L / pi[ia] * np.exp(rng.normal(0.0, mispricing, m_cp2)), Rb=L / pi[ib],
            fee=FEES[rng.integers(0, len(FEES), m_cp2)], ia=ia, ib=ib)
    if m_w2:
        ia, ib = _pairs(rng, n, m_w2, zipf_s)
        L = value(m_w2)
        wa = W2_CHOICES[rng.integers(0, len(W2_CHOICES), m_w2)]
        out["w2"] = dict(
            Ra=2 * wa * L / pi[ia] * np.exp(rng.normal(0.0, mispricing, m_w2)),
            Rb=2 * (1 - wa) * L / pi[ib],
            fee=FEES[rng.integers(0, len(FEES), m_w2)], wa=wa, ia=ia, ib=ib)
    if m_gn:
        lo, hi = gn_sizes
        sizes = rng.integers(lo, hi + 1, m_gn)
        gn = {}
        for k in range(lo, hi + 1):
            mk = int(np.sum(sizes == k))
            if mk == 0:
                continue
            # k distinct tokens per pool: draw, then redraw the (rare) rows holding a duplicate
            idx = rng.integers(0, n, size=(mk, k))
            while True:
                srt = np.sort(idx, axis=1)
                bad = np.any(srt[:, 1:] == srt[:, :-1], axis=1)
                if not bad.any():
                    break
                idx[bad] = rng.integers(0, n, size=(int(bad.sum()), k))
            idx = np.ascontiguousarray(idx.T).astype(np.int32)
            descending = rng.integers(0, 2, mk).astype(bool)
            w = np.where(descending[None, :], np.arange(k, 0, -1.0)[:, None], 1.0)
            w = w / w.sum(axis=0, keepdims=True)
            L = value(mk)
            R = k * w * L[None, :] / pi[idx]
            R[0] *= np.exp(rng.normal(0.0, mispricing, mk))
            gn[k] = dict(R=R, w=w, idx=idx, fee=FEES[rng.integers(0, len(FEES), mk)])
        out["gn"] = gn
    if m_pow2:
        # power-sum pools x^(1-t) + y^(1-t) (the generic bucket's tenant): marginal price p_a / p_b = (Rb / Ra)^t, so reserves
        # with Ra^t pi_a = Rb^t pi_b sit at the market; value L on the first leg, a mispricing like everyone else's
        ia, ib = _pairs(rng, n, m_pow2, zipf_s)
        t = np.array([0.2, 0.35, 0.5, 0.65, 0.8])[rng.integers(0, 5, m_pow2)]
        L = value(m_pow2)
        Ra = L / pi[ia]
        Rb = Ra * (pi[ia] / pi[ib]) ** (1.0 / t)
        Ra = Ra * np.exp(rng.normal(0.0, mispricing, m_pow2))
        out["pow2"] = dict(Ra=Ra, Rb=Rb, fee=FEES[rng.integers(0, len(FEES), m_pow2)], t=t, ia=ia, ib=ib)
    if m_curve2:
        # stable pairs: a Curve pool joins two tokens of one peg group (4 consecutive token ids
        # share a latent price up to 0.2 %, see `peg` below), so it sits near its 1:1 point.
        ia = rng.integers(0, n, size=m_curve2)
        ib = (ia // 4) * 4 + (ia % 4 + rng.integers(1, 4, size=m_curve2)) % 4
        ib = np.minimum(ib, n - 1)
        ib = np.where(ib == ia, (ia // 4) * 4, ib)
        ib = np.where(ib == ia, np.maximum(ia - 1, 0), ib)      # (n = 4q + 1: the last token is a peg group of its own -- pair it with its neighbour)
        ia = ia.astype(np.int32); ib = ib.astype(np.int32)
        L = value(m_curve2)
        Ra = L / pi[ia] * np.exp(rng.normal(0.0, mispricing, m_curve2))
        Rb = L / pi[ib]
        A = np.array([10.0, 50.0, 100.0, 200.0])[rng.integers(0, 4, m_curve2)]
        out["curve2"] = dict(Ra=Ra, Rb=Rb, fee=FEES[rng.integers(0, len(FEES), m_curve2)],
                             alpha=curve_alpha_from_A(Ra, Rb, A), ia=ia, ib=ib)
    if m_gk_stable or m_gk_sum:
        gk = {}
        lo, hi = gk_sizes
        for kind, m_all in (("stable", m_gk_stable), ("sum", m_gk_sum)):
            if not m_all:
                continue
            sizes = rng.integers(lo, hi + 1, m_all)
            for k in range(lo, hi + 1):
                mk = int(np.sum(sizes == k))
                if mk == 0:
                    continue
                if kind == "stable":           # k of the `peg` tokens of one peg group, in random order
                    if k > peg:
                        raise ValueError(f"synthetic stableswap table pools: at most {peg} tokens of one peg group (peg=...)")
                    grp = rng.integers(0, max(1, n // peg), mk) * peg
                    perm = np.argsort(rng.random((mk, peg)), axis=1)[:, :k]
                    idx = np.minimum(grp[:, None] + perm, n - 1)
                else:                          # k distinct tokens anywhere
                    idx = rng.integers(0, n, size=(mk, k))
                while True:
                    srt = np.sort(idx, axis=1)
                    bad = np.any(srt[:, 1:] == srt[:, :-1], axis=1)
                    if not bad.any():
                        break
                    idx[bad] = rng.integers(0, n, size=(int(bad.sum()), k))
                idx = np.ascontiguousarray(idx.T).astype(np.int32)
                L = value(mk)
                R = L[None, :] / pi[idx] * np.exp(rng.normal(0.0, mispricing, (k, mk)))
                fee = FEES[rng.integers(0, len(FEES), mk)]
                if kind == "stable":
                    A = np.array([10.0, 50.0, 100.0, 200.0])[rng.integers(0, 4, mk)]
                    param = np.prod(R, axis=0) * R.mean(axis=0) / (2.0 * A)
                else:
                    param = np.zeros(mk)
                gk[(kind, k)] = dict(idx=idx, R=R, fee=fee, param=param)
        out["gk"] = gk
    return out


def curve_alpha_from_A(Ra, Rb, A, iters=64):
    """alpha for which  x + y - alpha/(xy) >= const  is the on-chain 2-coin StableSwap
    invariant with amplification A:  alpha = D^3/(16 A),  D from 4A(x+y)+D = 4AD + D^3/(4xy)."""
    Ra = np.asarray(Ra, float); Rb = np.asarray(Rb, float); A = np.asarray(A, float)
    S = Ra + Rb
    D = S.copy()
    for _ in range(iters):
        f = 4 * A * S + D - 4 * A * D - D ** 3 / (4 * Ra * Rb)
        df = 1 - 4 * A - 3 * D ** 2 / (4 * Ra * Rb)
        D = D - f / df
    return D ** 3 / (16 * A)


def config(name, seed=0, scale=1.0, pool_seed=None, zipf_s=None):
    """BASELINE.json configs 2-5 (config 1 is the shipped script).  zipf_s: SURVEY 8(d)'s hub-weighted stress variant of the
    token choice (Zipf(s) instead of uniform); "C4x4": the HBM-STREAMING variant 8(d)'s cache caveat asks for -- 4e7
    constant-product pools = 1.28 GB per evaluation, five times the 256 MiB Infinity Cache."""
    s = lambda m: max(1, int(round(m * scale)))
    import functools
    make_network = functools.partial(globals()["make_network"], pool_seed=pool_seed, zipf_s=zipf_s)
    if name == "C2":      # 1e4 constant-product pools, 100 tokens
        return make_network(100, m_cp2=s(10_000), seed=seed)
    if name == "C3":      # 1e6 mixed Uniswap-v2 + Balancer pools, 1000 tokens
        return make_network(1000, m_cp2=s(700_000), m_w2=s(200_000), m_gn=s(100_000), seed=seed)
    if name == "C4":      # 1e7 constant-product pools, 2000 tokens (sharded over 8 GPUs)
        return make_network(2000, m_cp2=s(10_000_000), seed=seed)
    if name == "C4x4":    # 4e7 constant-product pools, 2000 tokens: 1.28 GB of pool columns per evaluation -- must stream from HBM
        return make_network(2000, m_cp2=s(40_000_000), seed=seed)
    if name == "C4shard":  # one GPU's share of C4
        return make_network(2000, m_cp2=s(1_250_000), seed=seed)
    if name == "GK":      # the K-asset table: n-asset stableswap and constant-sum pools among constant-product ones (tests)
        return make_network(200, m_cp2=s(20_000), m_gn=s(2_000), m_gk_stable=s(4_000), m_gk_sum=s(1_000), seed=seed)
    if name == "G4":      # the generic bucket: power-sum pools among constant-product and weighted ones (tests; not a BASELINE config)
        return make_network(200, m_cp2=s(20_000), m_w2=s(5_000), m_gn=s(3_000), m_pow2=s(20_000), seed=seed)
    if name == "C5":      # 5e5 Curve pools + basket liquidation
        return make_network(1000, m_cp2=s(50_000), m_curve2=s(500_000), seed=seed)
    raise ValueError(name)
