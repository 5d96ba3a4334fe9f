"""worker of tests/test_distributed.py: one rank of a world_size-N gloo job on CPU.

Runs the pool-sharded outer loop (this rank's shard evaluated by the C oracle, ONE all-reduce of
[psi | sum arb | diag] per dual evaluation, identical step on every rank) -- the same structure
libcfmm_hip.so runs with RCCL -- and writes what it found to argv[1]-<rank>.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import cfmm  # noqa: E402
from cfmm import synthetic  # noqa: E402
from oracle.c_oracle import Oracle  # noqa: E402


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert cfmm.distributed.env_world()[0] == rank and cfmm.distributed.env_world()[2] == world
    uid = cfmm.distributed.broadcast_unique_id(dist, lambda: bytes(range(128)))      # stand-in for ncclGetUniqueId
    net = synthetic.config("C3", scale=0.01, seed=3)
    part = cfmm.distributed.rank_network(net, rank, world)
    o = Oracle(net["n_tokens"]); o.add_network(part); o.set_utility(net["c"])
    calls = [0]

    def allreduce(buf):
        t = torch.from_numpy(buf)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        calls[0] += 1

    r = o.solve_sharded(net["c"], allreduce, tol=1e-7)
    res = dict(rank=rank, world=world, uid_ok=(uid == bytes(range(128))), pools=cfmm.problem.network_pool_count(part),
               evals=r["evals"], allreduces=calls[0], status=r["status"], primal=r["primal_value"], dual=r["dual_value"],
               gap=r["gap"], infeas=r["infeas"], nu=r["nu"].tolist())
    with open(f"{out}-{rank}.json", "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
