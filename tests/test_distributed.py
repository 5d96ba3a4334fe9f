"""CPU suite, part 3: the N > 1 path.  world_size-2 (and 3) gloo jobs run the product's pool-sharded host path
(cfmm.distributed.sharded_problem -> cfmm.Problem.solve: global decisions, start-price broadcast, all-gathered
constant-sum ties) with the C oracle standing in for the device and torch.distributed(gloo) for RCCL; the sharded
run must reproduce the unsharded solve: one all-reduce per dual evaluation, identical prices on every rank, same
optimum -- for the linear, liquidation and swap utilities and for the shipped scripts with their kinks."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cfmm
from cfmm import synthetic
from oracle.c_oracle import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


SCENARIOS = {2: ["arbitrage", "liquidate", "liquidation_py", "two_asset_py"], 3: ["arbitrage", "swap", "arbitrage_py"]}


def _run_world(world, tmp_path, scenarios=None):
    port = _free_port()
    out = str(tmp_path / "res")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", PYTHONDONTWRITEBYTECODE="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), out, ",".join(scenarios or SCENARIOS[world])], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, o.decode()[-2000:]
    return [json.load(open(f"{out}-{r}.json")) for r in range(world)]


def _unsharded(net, util, tol):
    from oracle_ctx import OracleContext
    p = cfmm.Problem.from_network(net, utility=util)
    p.ctx = OracleContext(net["n_tokens"])
    v = p.solve(tol=tol)
    return p, v


@pytest.mark.parametrize("world", [2, 3])
def test_pool_sharded_solve_over_gloo_matches_unsharded(oracle_lib, tmp_path, world):
    """the PRODUCT's sharded host path (cfmm.distributed.sharded_problem -> Problem.solve) under world_size 2 / 3"""
    import dist_worker
    from helpers import utility_of, golden
    from oracle import instances as I
    res = _run_world(world, tmp_path)
    net = synthetic.config("C3", scale=0.01, seed=3)
    for r in res:
        assert r["uid_ok"] and r["world"] == world and r["mismatch_caught"] and r["utility_mismatch_caught"]
    # 1. linear-utility arbitrage and the two basket utilities (start prices are a broadcast guess there)
    for key, util in (("arbitrage", cfmm.Arbitrage(net["c"])), ("liquidate", dist_worker.basket(net, "liquidate")),
                      ("swap", dist_worker.basket(net, "swap"))):
        if key not in SCENARIOS[world]:
            continue
        ref, v = _unsharded(net, util, 1e-6)
        assert ref.status == "optimal"
        assert sum(r[key]["pools"] for r in res) == cfmm.problem.network_pool_count(net)
        for r in res:
            q = r[key]
            assert q["status"] == "optimal" and q["gap"] <= 1e-6 and q["infeas"] <= 1e-6
            assert q["allreduces"] == q["evals"]                      # ONE collective per dual evaluation
            assert abs(q["value"] - v) <= 2e-6 * abs(v)
        # every rank started from bit-identical prices and took the identical sequence of steps
        for r in res[1:]:
            assert r[key]["nu0"] == res[0][key]["nu0"]
            assert r[key]["evals"] == res[0][key]["evals"]
            assert r[key]["nu"] == res[0][key]["nu"] and r[key]["psi"] == res[0][key]["psi"]
    # 2. the shipped scripts pool-sharded: a partially filled constant-sum pool held by one rank only
    g = golden()
    for key, gname in (("arbitrage_py", "arbitrage"), ("liquidation_py", "liquidation"), ("two_asset_py", "two_asset_10")):
        if key not in SCENARIOS[world]:
            continue
        want = g[gname]["survey"]["value"]
        for r in res:
            q = r[key]
            assert q["status"] == "optimal", (key, q)
            assert abs(q["value"] - want) <= 1e-8 * max(1.0, abs(want))
            assert len(q["theta"]) == 1 and 1e-6 < q["theta"][0][1] < 1 - 1e-6     # the kink was found and filled
        for r in res[1:]:
            assert r[key]["solves"] == res[0][key]["solves"] and r[key]["allreduces"] == res[0][key]["allreduces"]
            assert r[key]["nu"] == res[0][key]["nu"] and r[key]["theta"] == res[0][key]["theta"]


def test_auto_allreduce_trusts_the_one_shot_exchange_only_after_it_reproduced_the_reference(oracle_lib, tmp_path):
    """cfmm.distributed.attach_oneshot_checked (allreduce="auto", what bench.py --gpus N uses): all ranks switch to the
    one-shot exchange iff it reproduced the reference collective on every rank; a rank whose sums differ, or that could not
    map its peers' mailboxes, sends EVERY rank back to the reference path"""
    res = _run_world(2, tmp_path, ["auto_allreduce"])
    for r in res:
        a = r["auto_allreduce"]
        assert a["good"]["how"] == "oneshot" and a["good"]["enabled"] is True
        assert a["corrupt"]["how"] == "rccl" and a["corrupt"]["enabled"] is False
        assert a["refuse"]["how"] == "rccl" and a["refuse"]["enabled"] is False
    assert any("did not reproduce" in r["auto_allreduce"]["corrupt"]["note"] for r in res)
    assert any("hipIpcOpenMemHandle" in r["auto_allreduce"]["refuse"]["note"] for r in res)


def test_virtual_shards_sum_to_the_unsharded_evaluation(oracle_lib):
    """S shards evaluated one after the other and summed on the host equal the
    unsharded dual evaluation (partitioning + reduction logic, no collective)"""
    net = synthetic.config("C3", scale=0.02, seed=1)
    nu = net["c"] * np.exp(np.random.default_rng(0).normal(0, 0.02, net["n_tokens"]))
    o = Oracle(net["n_tokens"]); o.add_network(net); o.set_utility(net["c"])
    f, psi = o.eval(nu)
    for S in (2, 5, 8):
        fs, ps = 0.0, np.zeros_like(psi)
        for r in range(S):
            part = cfmm.distributed.rank_network(net, r, S)
            oo = Oracle(net["n_tokens"]); oo.add_network(part); oo.set_utility(net["c"])
            fr, pr = oo.eval(nu)
            fs += fr; ps += pr
        assert abs(fs - f) <= 1e-12 * abs(f)
        assert np.abs(ps - psi).max() <= 1e-12 * np.abs(psi).max()
