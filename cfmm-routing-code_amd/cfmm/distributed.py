"""Pool-sharding over the GPUs of one node: one process per GPU.

Given nu every pool subproblem is independent (the pools couple only through psi,
/root/reference/arbitrage.py:54), so rank r holds a contiguous slice of every SoA bucket, tokens /
prices / utility are replicated, and the ONLY exchange is one all-reduce of [psi | sum arb] per
dual evaluation.  On the GPUs that all-reduce is RCCL, enqueued by libcfmm_hip.so on its own stream
inside the captured outer iteration (cfmm_comm_init, include/cfmm.h); torch.distributed is used
only to agree on the 128-byte ncclUniqueId, for barriers and for the max-over-ranks clock.
"""
import os

from .problem import Problem, HostComm, shard_network


def env_world():
    """(rank, local_rank, world) from the torch.distributed.run environment (1 process: 0, 0, 1)"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def rank_network(net, rank, world):
    """this rank's shard of a bucketed network: contiguous equal-count slices of every bucket"""
    return shard_network(net, rank, world)


def broadcast_unique_id(dist, make_id, src=0):
    """rank `src` makes the communicator id (cfmm._lib.comm_unique_id), everyone receives it"""
    box = [make_id() if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    uid = bytes(box[0])
    if len(uid) != 128:
        raise ValueError(f"communicator id has {len(uid)} bytes, expected 128")
    return uid


def attach_oneshot(prob, dist):
    """switch this rank's collectives to the one-shot xGMI exchange (csrc/oneshot.hpp): every rank exports its mailbox as
    an IPC handle, the handles travel over torch.distributed, every rank maps its peers' mailboxes.  RCCL stays
    initialised underneath (it carries what does not fit a mailbox, e.g. the second-order method's Hessian)."""
    ctx = prob._ensure_ctx()
    mine = ctx.oneshot_export()
    box = [None] * dist.get_world_size()
    dist.all_gather_object(box, mine)
    ctx.oneshot_import(dist.get_world_size(), dist.get_rank(), box)


def sharded_problem(net, utility, dist=None, device=None, shard=True, context=None, allreduce=None):
    """Problem over this rank's shard, with the library's RCCL communicator initialised.

    net: the FULL network (shard=True: it is sliced here) or this rank's own pools (shard=False,
    e.g. bench.py's weak-scaling shards).  dist: an initialised torch.distributed (nccl) module.
    Host-side decisions of the solve (start prices, method, constant-sum ties) are then taken on global
    quantities through problem.HostComm, so that every rank issues the same device collectives.
    `context`: a ready device context to use instead of creating one on `device` (the CPU tests pass a stand-in
    whose collective is gloo; the product never does).  `allreduce`: "rccl" (default) or "oneshot" (also selected by
    CFMM_ALLREDUCE=oneshot): the per-evaluation all-reduce as ONE xGMI hop through peer-mapped mailboxes."""
    rank, local_rank, world = env_world()
    if dist is not None:
        rank, world = dist.get_rank(), dist.get_world_size()
    part = rank_network(net, rank, world) if (shard and world > 1) else net
    prob = Problem.from_network(part, utility=utility, device=local_rank if device is None else device)
    if context is not None:
        prob.ctx = context
    if dist is not None:                 # (a process group of one rank runs the same path)
        prob._host = HostComm(dist)
        prob._ensure_ctx()
        if context is None:
            from . import _lib
            prob.init_comm(world, rank, broadcast_unique_id(dist, _lib.comm_unique_id))
            how = (allreduce or os.environ.get("CFMM_ALLREDUCE", "rccl")).lower()
            if how not in ("rccl", "oneshot"):
                raise ValueError(f"allreduce={how!r}: expected 'rccl' or 'oneshot'")
            if how == "oneshot":
                attach_oneshot(prob, dist)
    return prob
