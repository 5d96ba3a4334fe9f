#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--no-cpu]

One *step* = one complete solve of the hot path (dual-decomposition routing, projected L-BFGS on
log-prices, on device) from a cold start to the 1e-6 certificates on one batch of synthetic input:
BASELINE config 3 (1e6 mixed Uniswap-v2 + Balancer pools / 1000 tokens, linear-utility
arbitrage) PER GPU -- the configuration the metric is quoted on.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank holds its own 1e6-pool shard of an N x 1e6
network over the same tokens (weak scaling) and the library all-reduces [psi | sum arb] over
RCCL once per dual evaluation.  value = pool-subproblems solved by all ranks / wall time of the
K timed solves (inputs resident in HBM; upload excluded).

Also reported on the same JSON line:
  roofline     -- the dominant evaluation kernel timed live with HIP events on the library's stream:
                  algorithmic bytes per launch / average launch duration vs the 8 TB/s HBM peak
  cpu_baseline -- oracle/cfmm_oracle.c (the CPU restatement, OpenMP on all host cores) running the
                  same solve on the same instance, timed in this run (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (~6.3 TB/s achievable)
# HBM bytes per eval_kernel launch on the C3 workload from the rocprofv3 PMC passes committed under
# profiles/ (FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM, + WRITE_SIZE); None until measured
TRAFFIC_BYTES = (2 * 22898 + 2008) * 1024      # profiles/r01c_rocprofv3_pmc_medians.csv, config C3
TRAFFIC_NOTE = ("C3's 43.4 MB working set fits the 256 MiB Infinity Cache: back-to-back launches are cache-served; "
                "profiles/ holds the PMC traffic and the HBM-streaming (>= 1e7 pools) variant")

BYTES_PER_POOL = {"cp2": 32, "w2": 40, "sum2": 32, "curve2": 40}     # SURVEY 8(d); k-asset: 20 + 20 k (+ log fee)
KIND_ID = {"cp2": 0, "w2": 1, "sum2": 2, "curve2": 3}


def kernel_table(prob, reps):
    """the fused evaluation kernel timed live with HIP events on the library's stream: row 0 is the
    launch one dual evaluation makes (every bucket); the other rows restrict it to one bucket"""
    from cfmm import _lib
    net = prob.net
    parts = []
    for key in ("cp2", "w2", "sum2", "curve2"):
        if key in net:
            m = len(net[key]["Ra"])
            parts.append((f"eval_kernel[{key} only]", KIND_ID[key], m, m * BYTES_PER_POOL[key]))
    for k, b in sorted(net.get("gn", {}).items()):
        m = b["R"].shape[1]
        parts.append((f"eval_kernel[gn{k} only]", -k, m, m * (20 + 20 * k)))
    rows = [dict(kernel="eval_kernel", pools=sum(p[2] for p in parts), bytes=sum(p[3] for p in parts),
                 seconds=prob.ctx.time_eval_kernel(_lib.TIME_ALL, reps))]
    for name, code, m, nbytes in parts:
        rows.append(dict(kernel=name, pools=m, bytes=nbytes, seconds=prob.ctx.time_eval_kernel(code, reps)))
    for r in rows:
        r["GBps"] = r["bytes"] / r["seconds"] / 1e9
        r["pools_per_s"] = r["pools"] / r["seconds"]
        r["us"] = r["seconds"] * 1e6
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-solves", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--kernel-reps", type=int, default=50)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the pool-sharded code path (process group, RCCL communicator, all-reduce per evaluation) even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
        args.gpus = world

    import cfmm
    from cfmm import synthetic, _lib

    dist = None
    sharded = world > 1 or args.force_dist
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    net = synthetic.config(args.config, seed=0, scale=args.scale, pool_seed=(rank if world > 1 else None))
    # weak scaling: every rank generates its OWN 1e6-pool shard (same tokens / prices / utility)
    prob = cfmm.distributed.sharded_problem(net, cfmm.Arbitrage(net["c"]), dist=dist, device=local_rank, shard=False)
    prob._ensure_ctx()

    def sync():
        if sharded:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        prob.solve(tol=args.tol)
    sync()
    evals = 0
    dev_s = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        prob.solve(tol=args.tol)
        evals += prob.stats["evals"]
        dev_s += prob.stats["device_seconds"]
        if prob.status != "optimal":
            raise SystemExit(f"rank {rank}: solve ended with status {prob.status} (gap {prob.gap:.2e}, infeas {prob.infeas:.2e})")
    sync()
    dt = time.perf_counter() - t0
    if sharded:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    subproblems = evals * prob.m * world       # every rank runs the same number of evaluations
    value = subproblems / dt

    out = None
    if rank == 0:
        rows = kernel_table(prob, args.kernel_reps)
        dom = rows[0]
        mix = ", ".join(f"{k}={len(net[k]['Ra'])}" for k in ("cp2", "w2", "curve2", "sum2") if k in net)
        if "gn" in net:
            mix += ", gn3-8=" + str(sum(b["R"].shape[1] for b in net["gn"].values()))
        out = {
            "metric": "pool-subproblems/sec to 1e-6 rel-gap; 1e6 pools / 1k tokens; 1/2/4/8 GPU",
            "value": value, "unit": "pool-subproblems/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {prob.m} pools per GPU ({mix}) / {net['n_tokens']} tokens, "
                                   f"linear-utility arbitrage, cold-start solve to gap,infeas <= {args.tol:g}",
                       "pools_per_gpu": prob.m, "tokens": net["n_tokens"], "seed": 0,
                       "parallelism": f"pool-sharding x{world}" if world > 1 else "single GPU"},
            "evals_per_solve": evals / args.steps,
            "device_ms_per_step": 1e3 * dev_s / args.steps,
            "us_per_eval": 1e6 * dt / max(evals, 1),
            "gap": prob.gap, "infeas": prob.infeas, "objective": prob.value,
            "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["GBps"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": dom["GBps"] / HBM_PEAK_GBS, "traffic": TRAFFIC_BYTES,
                         "algorithmic_bytes_per_launch": dom["bytes"], "avg_launch_us": dom["seconds"] * 1e6,
                         "note": TRAFFIC_NOTE,
                         "all_kernels": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]},
        }
        if world == 1 and not args.no_cpu:
            from oracle.c_oracle import Oracle
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            # calibrate the thread count on single dual evaluations (a cgroup quota can make "all logical
            # CPUs" the slowest choice), then time full solves with the best one
            best = None
            for th in sorted({avail, max(1, avail // 2), max(1, avail // 4), min(avail, 64), min(avail, 32), min(avail, 16)}, reverse=True):
                o = Oracle(net["n_tokens"], threads=th)
                o.add_network(net); o.set_utility(net["c"])
                o.eval(net["c"])
                t0 = time.perf_counter()
                for _ in range(3):
                    o.eval(net["c"])
                dt1 = (time.perf_counter() - t0) / 3
                if best is None or dt1 < best[1]:
                    best = (th, dt1, o)
            cores, eval_s, o = best
            o.solve(net["c"], tol=args.tol)          # warm the OpenMP pool
            t0 = time.perf_counter(); ce = 0; ns = 0
            while ns < args.cpu_solves or time.perf_counter() - t0 < args.cpu_seconds:     # >= 10 s: past any cgroup burst allowance
                r = o.solve(net["c"], tol=args.tol); ce += r["evals"]; ns += 1
            cdt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": ce * prob.m / cdt, "unit": "pool-subproblems/s", "cores": cores, "kind": "port",
                                   "sample": f"{ns} full solves of the same {prob.m}-pool instance to the same tolerance by "
                                             f"oracle/cfmm_oracle.c (OpenMP, {cores} threads: the fastest of the thread counts tried on "
                                             f"the {avail} CPUs this process may use), {cdt:.1f} s; cvxpy (the reference's solver stack) "
                                             "is not installed in this image",
                                   "evals_per_solve": ce / ns, "objective": r["primal_value"],
                                   "single_evaluation_ms": eval_s * 1e3}
        print(json.dumps(out))
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
