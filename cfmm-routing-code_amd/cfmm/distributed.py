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
    try:
        ctx.oneshot_import(dist.get_world_size(), dist.get_rank(), box)
    finally:
        # cfmm_oneshot_attach clears this rank's mailbox once more after the (slow) handle imports: no peer may start an
        # exchange -- and raise its flag in that mailbox -- before EVERY rank has finished attaching (a rank whose
        # import failed still meets the others here: the collectives of all ranks stay matched)
        dist.barrier()


def attach_oneshot_checked(prob, dist, evaluations=4, solve_evals=18):
    """allreduce="auto": set the one-shot exchange up and TRUST it only after it has reproduced RCCL on the real peers --
    the same sharded dual evaluations (several in a row: both parity slots, epochs running ahead) and the same short
    solve, compared on every rank; any rank's failure (IPC mapping refused, a timed-out exchange, a differing sum) sends
    ALL ranks back to RCCL.  Returns ("oneshot" | "rccl", reason)."""
    import numpy as np
    from ._lib import CfmmError, METHODS
    host = prob._host
    ctx = prob._ensure_ctx()
    why = ""
    try:
        attach_oneshot(prob, dist)               # (leaves the one-shot path enabled)
        ctx.oneshot_enable(False)
    except CfmmError as e:
        why = f"rank {host.rank}: {e}"
    failed = host.allreduce_sum(np.array([1.0 if why else 0.0]))[0]
    if failed:
        return "rccl", why or "a peer could not map the mailboxes"
    u = prob.utility
    base = np.where(u.c > 0, u.c, 1.0)
    pts = [base * np.exp(0.01 * np.sin(1.0 + k + np.arange(prob.n))) for k in range(evaluations)]
    prob._send_utility()                         # (cfmm_solve refuses a context without a utility; sharded_problem only uploaded pools)

    def run():
        out = [prob.eval_dual(nu) for nu in pts]
        st = ctx.solve(pts[0], tol=1e-6, max_evals=solve_evals, method=METHODS["lbfgs"])
        nu, psi = ctx.get_solution()
        return out, (st["evals"], st["dual_value"], nu, psi)

    ok = True
    try:
        ref, ref_solve = run()                   # RCCL
        ctx.oneshot_enable(True)
        got, got_solve = run()
        for (f0, p0), (f1, p1) in zip(ref, got):
            ok &= bool(np.isfinite(f1) and abs(f1 - f0) <= 1e-10 * max(1.0, abs(f0)) and
                       np.all(np.isfinite(p1)) and np.abs(p1 - p0).max() <= 1e-10 * max(1.0, np.abs(p0).max()))
        ok &= bool(got_solve[0] == ref_solve[0] and abs(got_solve[1] - ref_solve[1]) <= 1e-8 * max(1.0, abs(ref_solve[1])) and
                   np.abs(got_solve[2] - ref_solve[2]).max() <= 1e-6 * np.abs(ref_solve[2]).max())
        if not ok:
            why = f"rank {host.rank}: the one-shot exchange did not reproduce RCCL"
    except CfmmError as e:
        ok, why = False, f"rank {host.rank}: {e}"
    failed = host.allreduce_sum(np.array([0.0 if ok else 1.0]))[0]
    if failed:
        ctx.oneshot_enable(False)
        return "rccl", why or "a peer's check failed"
    return "oneshot", f"reproduced RCCL on {evaluations} evaluations and a {solve_evals}-evaluation solve"


def sharded_problem(net, utility, dist=None, device=None, shard=True, context=None, allreduce=None, rccl=True):
    """Problem over this rank's shard, with the library's RCCL communicator initialised.

    net: the FULL network (shard=True: it is sliced here) or this rank's own pools (shard=False,
    e.g. bench.py's weak-scaling shards).  dist: an initialised torch.distributed (nccl) module.
    Host-side decisions of the solve (start prices, method, constant-sum ties) are then taken on global
    quantities through problem.HostComm, so that every rank issues the same device collectives.
    `context`: a ready device context to use instead of creating one on `device` (the CPU tests pass a stand-in
    whose collective is gloo; the product never does).  `allreduce`: "rccl" (default), "oneshot" (also selected by
    CFMM_ALLREDUCE=oneshot): the per-evaluation all-reduce as ONE xGMI hop through peer-mapped mailboxes, or "auto": the
    one-shot exchange if -- and only if -- it reproduces RCCL on these peers at start-up (attach_oneshot_checked); what
    was chosen, and why, is left in prob.allreduce / prob.allreduce_note.  `rccl=False` (with allreduce="oneshot"): no
    RCCL communicator at all -- the ranks may then share ONE GPU (RCCL refuses that), which is how the IPC path of the
    one-shot exchange is tested between real processes on a one-GPU box."""
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
            how = (allreduce or os.environ.get("CFMM_ALLREDUCE", "rccl")).lower()
            if rccl:
                prob.init_comm(world, rank, broadcast_unique_id(dist, _lib.comm_unique_id))
            elif how != "oneshot":
                raise ValueError("rccl=False needs allreduce='oneshot'")
            if how not in ("rccl", "oneshot", "auto"):
                raise ValueError(f"allreduce={how!r}: expected 'rccl', 'oneshot' or 'auto'")
            prob.allreduce, prob.allreduce_note = how, ""
            if how == "oneshot":
                attach_oneshot(prob, dist)
            elif how == "auto" and world > 1:
                prob.allreduce, prob.allreduce_note = attach_oneshot_checked(prob, dist)
            elif how == "auto":
                prob.allreduce = "rccl"
    return prob
