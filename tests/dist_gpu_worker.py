"""worker of tests/test_gpu.py::test_one_shot_all_reduce_against_rccl_on_real_peers: one rank per GPU (torch.distributed.run).
Each rank holds one pool shard; the same evaluations and the same solve run once over RCCL and once over the one-shot
xGMI mailboxes (csrc/oneshot.hpp).  Rank 0 writes the comparison to argv[1]."""
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


def same_gpu():
    """argv[2] == "same_gpu": the ranks are PROCESSES sharing GPU 0 (gloo for the host side, no RCCL -- it refuses two
    ranks on one device): the real IPC path of the one-shot exchange (hipIpcGetMemHandle / OpenMemHandle, mailboxes
    written by another process, system-scope visibility) without a second GPU"""
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    net = synthetic.config("C3", scale=0.1, seed=4)
    n = net["n_tokens"]
    nus = [net["c"] * np.exp(np.random.default_rng(3 + k).normal(0, 0.02, n)) for k in range(3)]
    res = {}
    for det in (False, True):
        p = cfmm.distributed.sharded_problem(net, cfmm.Arbitrage(net["c"]), dist=dist, device=0, allreduce="oneshot", rccl=False)
        if det:
            p.ctx.set_deterministic(True)
        ev = [p.eval_dual(nu) for nu in nus]
        v = p.solve(tol=1e-6, method="lbfgs")          # (first order only: the second-order fall-back all-reduces a Hessian, which needs RCCL)
        mine = dict(f=[e[0] for e in ev], psi=[e[1].tolist() for e in ev], value=v, evals=p.stats["evals"], status=p.status,
                    nu=p.nu.tolist(), ranks=p.stats["n_ranks"])
        box = [None] * world
        dist.all_gather_object(box, mine)
        ref = None
        if rank == 0:                                   # the unsharded problem, same GPU, after the ranks are done
            q = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]), deterministic=det)
            ev0 = [q.eval_dual(nu) for nu in nus]
            ref = dict(f=[e[0] for e in ev0], psi=[e[1].tolist() for e in ev0], value=q.solve(tol=1e-6, method="lbfgs"), evals=q.stats["evals"], nu=q.nu.tolist())
            q.close()
        res["det" if det else "fp64"] = dict(ranks=box, unsharded=ref)
        p.close()
        dist.barrier()
    if rank == 0:
        with open(sys.argv[1], "w") as fh:
            json.dump(dict(world=world, res=res), fh)
    dist.barrier()
    dist.destroy_process_group()


def same_gpu_gk():
    """argv[2] == "same_gpu_gk": 1 000 K-asset constant-sum pools (arbitrage.py:73-74 over 3-5 tokens) among 20 000 constant-product
    pools, pool-sharded over processes sharing GPU 0.  Their optimum sits on kinks of both kinds (a leg partially drained, two
    tokens tied for cheapest): the host's active-set loop must take the SAME decisions on every rank from records that carry the
    owning rank's pool -- round 5 looked the pool up in the local shard (ranks desynchronised or IndexError)."""
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    net = synthetic.make_network(200, m_cp2=20000, m_gk_sum=1000, seed=3)
    p = cfmm.distributed.sharded_problem(net, cfmm.Arbitrage(net["c"]), dist=dist, device=0, allreduce="oneshot", rccl=False)
    v = p.solve(tol=1e-6, max_evals=1500, method="lbfgs")
    mine = dict(value=v, status=p.status, gap=p.gap, infeas=p.infeas, evals=p.stats["evals"], rounds=p.stats.get("rounds"), nu=p.nu.tolist(), psi=p.psi.tolist(),
                theta=sorted([list(k), th] for k, (_, th) in p._theta.items()), owners=sorted({k[0] for k in p._theta}))
    box = [None] * world
    dist.all_gather_object(box, mine)
    ref = None
    if rank == 0:
        q = cfmm.Problem.from_network(net, utility=cfmm.Arbitrage(net["c"]))
        ref = dict(value=q.solve(tol=1e-6, max_evals=1500, method="lbfgs"), status=q.status, ntheta=len(q._theta))
        q.close()
    p.close()
    dist.barrier()
    if rank == 0:
        with open(sys.argv[1], "w") as fh:
            json.dump(dict(world=world, ranks=box, unsharded=ref), fh)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "same_gpu":
        return same_gpu()
    if len(sys.argv) > 2 and sys.argv[2] == "same_gpu_gk":
        return same_gpu_gk()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    rank, world = dist.get_rank(), dist.get_world_size()
    net = synthetic.config("C3", scale=0.2, seed=4)
    n = net["n_tokens"]
    nu = net["c"] * np.exp(np.random.default_rng(3).normal(0, 0.02, n))
    res = {}
    for how in ("rccl", "oneshot"):
        for det in (False, True):
            p = cfmm.distributed.sharded_problem(net, cfmm.Arbitrage(net["c"]), dist=dist, device=local, allreduce=how)
            if det:
                p.ctx.set_deterministic(True)
            f, psi = p.eval_dual(nu)
            v = p.solve(tol=1e-7)
            # every rank must hold the same bits: compare with rank 0's
            ref = [None]
            if rank == 0:
                ref[0] = (f, psi.tolist(), p.nu.tolist(), p.stats["evals"])
            dist.broadcast_object_list(ref, src=0)
            same = (f == ref[0][0]) and psi.tolist() == ref[0][1] and p.nu.tolist() == ref[0][2] and p.stats["evals"] == ref[0][3]
            flags = [None] * world
            dist.all_gather_object(flags, bool(same))
            res[f"{how}{'_det' if det else ''}"] = dict(f=f, psi=psi.tolist(), value=v, evals=p.stats["evals"], status=p.status,
                                                        nu=p.nu.tolist(), all_ranks_same_bits=all(flags))
            p.close()
    # allreduce="auto" (bench.py's default): on real peers the start-up check must pass and pick the one-shot exchange
    p = cfmm.distributed.sharded_problem(net, cfmm.Arbitrage(net["c"]), dist=dist, device=local, allreduce="auto")
    v = p.solve(tol=1e-7)
    auto = dict(how=p.allreduce, note=p.allreduce_note, value=v, status=p.status)
    p.close()
    if rank == 0:
        with open(sys.argv[1], "w") as fh:
            json.dump(dict(world=world, res=res, auto=auto), fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
