"""CPU suite, part 3: the N > 1 path.  world_size-2 (and 3) gloo jobs run the pool-sharded outer loop
with the C oracle standing in for the device and torch.distributed(gloo) for RCCL; the sharded run
must reproduce the unsharded solve: one all-reduce per dual evaluation, identical prices on every
rank, same optimum."""
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


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run_world(world, tmp_path):
    port = _free_port()
    out = str(tmp_path / "res")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", PYTHONDONTWRITEBYTECODE="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), out], env=env,
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


@pytest.mark.parametrize("world", [2, 3])
def test_pool_sharded_solve_over_gloo_matches_unsharded(oracle_lib, tmp_path, world):
    res = _run_world(world, tmp_path)
    net = synthetic.config("C3", scale=0.01, seed=3)
    o = Oracle(net["n_tokens"]); o.add_network(net); o.set_utility(net["c"])
    ref = o.solve(net["c"], tol=1e-7)
    assert ref["status"] == 1
    assert sum(r["pools"] for r in res) == cfmm.problem.network_pool_count(net)
    for r in res:
        assert r["uid_ok"] and r["world"] == world
        assert r["status"] == 1 and r["gap"] <= 1e-7 and r["infeas"] <= 1e-7
        assert r["allreduces"] == r["evals"]                      # ONE collective per dual evaluation
        assert abs(r["primal"] - ref["primal_value"]) <= 1e-7 * abs(ref["primal_value"])
        assert np.abs(np.asarray(r["nu"]) / ref["nu"] - 1).max() <= 1e-6
    # every rank took the identical sequence of steps (no broadcast of nu is ever needed)
    for r in res[1:]:
        assert r["evals"] == res[0]["evals"]
        assert np.array_equal(np.asarray(r["nu"]), np.asarray(res[0]["nu"]))


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
