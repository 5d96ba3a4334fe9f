"""CPU suite: host side of the K-asset table (csrc/phik.hpp) -- the cvx recogniser's mapping of k-asset stableswap and
constant-sum constraint lines onto the table's buckets (no device involved: the pattern match alone)."""
import numpy as np

import cfmm
import cfmm.cvx as cp


def test_k_asset_constraint_lines_are_mapped_onto_the_table_buckets():
    rng = np.random.default_rng(0)
    n = 5
    idx = [[0, 1, 2], [1, 3, 4, 0], [2, 4]]
    R = [np.array([3.0, 4.0, 5.0]), np.array([2.0, 2.5, 3.0, 3.5]), np.array([1.0, 2.0])]
    fee = [0.999, 0.997, 0.99]
    D = [cp.Variable(len(i), nonneg=True) for i in idx]
    L = [cp.Variable(len(i), nonneg=True) for i in idx]
    S = []
    for i in idx:
        A = np.zeros((n, len(i))); A[np.asarray(i), np.arange(len(i))] = 1.0; S.append(A)
    psi = cp.sum([A @ (l - d) for A, d, l in zip(S, D, L)])
    x = [r + g * d - l for r, g, d, l in zip(R, fee, D, L)]
    al = 7.0
    cons = [cp.sum(x[0]) - al * cp.inv_prod(x[0]) >= float(R[0].sum() - al / R[0].prod()),      # three-token stableswap
            cp.sum(x[1]) >= cp.sum(R[1]), x[1] >= 0,                                             # four-token constant sum
            cp.geo_mean(x[2]) >= cp.geo_mean(R[2]),
            psi >= 0]
    prob = cp.Problem(cp.Maximize(np.ones(n) @ psi), cons)
    pools, local, n_tok, util = prob._match()
    assert [p["kind"] for p in pools] == ["curve", "geomean", "sum"] or sorted(p["kind"] for p in pools) == ["curve", "geomean", "sum"]
    net, where = cfmm.pack(n_tok, local, [p["R"] for p in pools], [p["fee"] for p in pools], [p["kind"] for p in pools],
                           [p["w"] for p in pools], [p.get("param") for p in pools])
    assert set(net["gk"]) == {("stable", 3), ("sum", 4)} and net["gk"][("stable", 3)]["param"][0] == al
    assert "cp2" in net and cfmm.problem.network_pool_count(net) == 3
