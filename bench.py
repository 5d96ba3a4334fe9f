#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C2|C3|C4|C5] [--no-cpu] [--share-gpu]

One *step* = one complete solve of the hot path (dual-decomposition routing, projected L-BFGS on
log-prices, on device) from a cold start to the 1e-6 certificates on one batch of synthetic input.

  --config C3 (default; the configuration the metric is quoted on): BASELINE config 3 -- 1e6 mixed Uniswap-v2 +
      Balancer pools / 1000 tokens, linear-utility arbitrage -- PER GPU.  With N > 1 every rank holds its own
      1e6-pool shard of an N x 1e6 network over the same tokens: WEAK scaling.
  --config C4: BASELINE config 4 exactly -- 1e7 constant-product pools / 2000 tokens -- split N ways (1.25e6 pools
      per GPU at N = 8): STRONG scaling.  At N = 1 the whole 320 MB set streams from HBM on one GPU.

  --config C2: BASELINE config 2 -- 1e4 constant-product pools / 100 tokens, linear-utility arbitrage (launch-latency
      bound: 0.3 MB of pool data per evaluation).
  --config C5: BASELINE config 5 -- 5e5 stableswap + 5e4 constant-product pools / 1000 tokens, a 10-token basket
      liquidation (liquidation.py:57,77-80) -- through the second-order path (barrier-smoothed dual Newton); its
      dominant kernel is the dense factorisation of a Newton step, priced against the fp64 vector peak.
  --share-gpu (with --gpus N > 1 on a box with ONE GPU): the N ranks are N processes on device 0 -- gloo for the host
      side, no RCCL (it refuses two ranks on one device), the per-evaluation exchange through the hipIpc-mapped
      one-shot mailboxes: the whole multi-rank path of this file (self-launch, torch.distributed.run, pool shards,
      per-iteration split, max-over-ranks clock, one JSON line) end to end without a second GPU.  Not a scaling number.

With N > 1 there is one process per GPU and the library all-reduces [psi | sum arb] over RCCL once per dual
evaluation.  `python bench.py --gpus N` launches those processes itself (torch.distributed.run on 127.0.0.1) when it
is not already running under a launcher; `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
works as well.  value = pool-subproblems solved by all ranks / wall time of the K timed solves (barrier +
synchronize on both sides, max over ranks; inputs resident in HBM, upload excluded).

Also on the same JSON line:
  roofline     -- the dominant evaluation kernel timed live with HIP events on the library's stream:
                  algorithmic bytes per launch / average launch duration vs the 8 TB/s HBM peak; next to it the
                  rocprofv3 average and PMC traffic of the newest summary committed under profiles/
  per_iteration_us -- evaluation launch / accumulator fold / RCCL all-reduce / nu update + launch boundaries
  cpu_baseline -- oracle/cfmm_oracle.c (the CPU restatement, OpenMP on the host cores) running the
                  same solve on the same instance, timed in this run (rank 0, N = 1 only)
"""
import argparse
import csv
import glob
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "cfmm-routing-code_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (~6.3 TB/s achievable)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # AMD's MI355X figure for vector fp64 (256 CUs x 4 SIMDs x 16 fp64 FMA lanes x 2.4 GHz x 2)
SIMDS, CLOCK_HZ = 1024, 2.4e9  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz max clock
TRAFFIC_NOTE = {"C3": "C3's 43.4 MB working set fits the 256 MiB Infinity Cache: back-to-back launches are cache-served; "
                      "profiles/ holds the PMC traffic and the HBM-streaming (>= 1e7 pools) variant",
                "C4": "320 MB of algorithmic pool columns per launch at N = 1, 210 MB as stored (compact mirror) against a 256 MiB Infinity Cache: "
                      "PARTLY CACHE-SERVED, not an HBM-streaming figure -- consecutive launches walk the pools in opposite directions (ping-pong), "
                      "so most of a launch finds its data still in the Infinity Cache (FETCH_SIZE counts Infinity-Cache hits: MI355X_MICROARCH.md). "
                      "The honest HBM fraction is --config C4x4",
                "C4x4": "1.28 GB of algorithmic pool columns per launch (4e7 constant-product pools), 0.84 GB as stored (compact mirror: ids in one "
                        "word, the fee as a byte index), more than three times the Infinity Cache: this set MUST stream from HBM whatever the walk "
                        "direction -- the figure SURVEY 8(d)'s cache caveat asks for.  `frac` (algorithmic) can exceed what HBM delivers because "
                        "fewer bytes are moved; `hbm_frac` is the bandwidth statement",
                "C2": "0.32 MB of pool data per evaluation: launch-latency bound -- neither fraction says much",
                "C5": "second-order path: the dominant kernel group is the dense n x n Cholesky of a Newton step (a latency chain of "
                      "dependent launches, priced against the fp64 vector peak); the smoothed evaluation is fp64-issue / divergence bound"}

BYTES_PER_POOL = {"cp2": 32, "w2": 40, "sum2": 32, "curve2": 40}     # SURVEY 8(d); k-asset: 20 + 20 k (+ log fee)
KIND_ID = {"cp2": 0, "w2": 1, "sum2": 2, "curve2": 3}
METRIC = "pool-subproblems/sec to 1e-6 rel-gap; 1e6 pools / 1k tokens; 1/2/4/8 GPU"


def profile_record(config, tag=""):
    """HBM bytes per launch (PMC, corrected as MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE doubled on gfx950,
    + WRITE_SIZE) and the rocprofv3 kernel-trace average of the same launch, from the NEWEST summaries committed under
    profiles/ -- read at run time so that the bench line cannot go stale silently.  Two records: "iter" = iter_kernel
    (the one launch per outer iteration; averages over its FULL launches, the idle run-ahead launches behind the end of a
    solve excluded) from tools/profile_iter.py's trace, "eval" = eval_kernel alone from tools/profile_eval.py's."""
    import json
    out = {k: dict(traffic=None, traffic_file=None, rocprof_avg_us=None, rocprof_file=None, valu_busy=None, sq_busy=None, pmc_ns=None,
                   l2_hit_rate=None) for k in ("iter", "eval")}
    config = config + tag                              # (e.g. "C3" + "zipf": the stress variant's own profile rows)
    pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_pmc_medians.csv")))
    wanted = ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "DISPATCH_NS_of_the_SQ_BUSY_CYCLES_pass", "TCC_HIT_sum", "TCC_MISS_sum")
    for which, cfg, name in (("iter", config + "iter", "iter_kernel"), ("eval", config, "eval_kernel")):
        for f in reversed(pmc):
            rows = {}
            for r in csv.DictReader(open(f)):
                if r["config"] == cfg and name in r["kernel"] and r["counter"] in wanted:
                    rows.setdefault(r["kernel"], {})[r["counter"]] = float(r["median"])
            rows = {k: v for k, v in rows.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
            if rows:                                   # the instantiation that moves the most bytes: the dominant one
                v = max(rows.values(), key=lambda v: v["FETCH_SIZE"])
                out[which]["traffic"] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
                out[which]["traffic_file"] = os.path.relpath(f, ROOT)
                # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs: x 4 = SIMD-cycles the vector ALU was issuing;
                # SQ_BUSY_CYCLES of the SAME pass = shader-clock cycles with a wave resident, summed over the 32 shader engines
                # (all busy for the whole of these launches): / 32 = the dispatch's length in shader clocks -- their ratio is a
                # fraction in which neither a clock nor a live duration is assumed; with the dispatch's duration under that pass
                # it also gives the effective clock under the profiler (1.9-2.0 GHz, not the 2.4 GHz maximum).
                # (GRBM_GUI_ACTIVE, which round 3's verdict suggested, comes out at 27 counts per ns on this stack -- no whole
                #  number of 2 GHz domains -- and is recorded in the summaries but not used.)
                out[which]["valu_busy"] = 4.0 * v["SQ_ACTIVE_INST_VALU"] if "SQ_ACTIVE_INST_VALU" in v else None
                out[which]["sq_busy"] = v.get("SQ_BUSY_CYCLES")
                out[which]["pmc_ns"] = v.get("DISPATCH_NS_of_the_SQ_BUSY_CYCLES_pass")
                if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v and v["TCC_HIT_sum"] + v["TCC_MISS_sum"] > 0:
                    out[which]["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
                break
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_durations.json")))):
        d = json.load(open(f)).get(config + "iter", {})
        cand = [v for k, v in d.items() if "iter_kernel" in k]
        if cand:
            v = max(cand, key=lambda v: v["full_launches"])
            out["iter"]["rocprof_avg_us"] = v["full_mean_us"]
            out["iter"]["rocprof_file"] = os.path.relpath(f, ROOT)
            break
    stats = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_rocprofv3_kernel_stats_{config}.csv")))
    for f in reversed(stats):
        cand = [r for r in csv.DictReader(open(f)) if "eval_kernel" in r.get("Name", "")]
        if cand:
            r = max(cand, key=lambda r: float(r["TotalDurationNs"]))
            out["eval"]["rocprof_avg_us"] = float(r["AverageNs"]) / 1e3
            out["eval"]["rocprof_file"] = os.path.relpath(f, ROOT)
            break
    return out


def newton_profile_record(kernel):
    """the factorisation kernel's average launch out of the newest committed kernel-trace summary of the Newton path
    (profiles/r*_rocprofv3_kernel_stats_C5newton.csv, tools/gpu_profile.sh) -- the C5 line's rocprof cross-check"""
    for f in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_kernel_stats_C5newton.csv")))):
        cand = [r for r in csv.DictReader(open(f)) if kernel + "(" in r.get("Name", "") or r.get("Name", "").endswith(kernel)]
        if cand:
            r = max(cand, key=lambda r: float(r["TotalDurationNs"]))
            return float(r["AverageNs"]) / 1e3, int(r["Calls"]), os.path.relpath(f, ROOT)
    return None, None, None


def numpy_one_thread(net, budget_s=6.0):
    """BASELINE.md section 4, throughput baseline B: the dual evaluation as vectorised NumPy on ONE thread (oracle/pools_np.py: the
    restatements the C twin is pinned against; elementwise ufuncs and a bincount scatter -- nothing in it is multi-threaded), at the
    market prices, for at most ~budget_s seconds.  Returns pools per second of evaluation and what was timed."""
    from oracle import pools_np as P
    nu = net["c"]

    def one():
        psi, _, m = P.dual_eval_network(net, nu)
        return m, psi

    t0 = time.perf_counter(); m, _ = one(); first = time.perf_counter() - t0
    times = [first]
    while sum(times) + min(times) < budget_s and len(times) < 4:
        t0 = time.perf_counter(); one(); times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": m / best, "unit": "pool-subproblems/s", "cores": 1, "kind": "port",
            "sample": f"{len(times)} dual evaluations of the same network ({m} pools) by oracle/pools_np.py (vectorised NumPy, one thread), "
                      f"best {1e3 * best:.1f} ms per evaluation -- evaluations only, no outer iteration (BASELINE.md section 3's anchors: 1.3e7 /s at C3 on the survey box)"}


def clock_ghz(samples, i0=0, i1=None):
    """shader clock over samples [i0, i1) of the probe (cfmm_clock_probe_*: rows of {shader cycles, 100 MHz ticks}), GHz"""
    a = np.asarray(samples[i0:i1], dtype=np.float64)
    if len(a) < 2 or a[-1, 1] <= a[0, 1]:
        return None
    # the MEDIAN of the per-interval rates, not last-minus-first: one bench line of round 6 read 3.28 GHz end to end on a pass whose
    # re-runs all read 2.39-2.40 with every 100 us interval between 2.29 and 2.44 -- a single discontinuity in one of the two counters
    # (never reproduced) moves the end-to-end quotient and leaves the median where it is
    dc, dt = np.diff(a[:, 0]), np.diff(a[:, 1]) * 10.0
    ok = dt > 0
    return float(np.median(dc[ok] / dt[ok])) if ok.any() else None


def device_state():
    """what the driver exposes about the device's partitioning / clock / power state to an ordinary user (sysfs; every entry optional):
    recorded so that a slow lease can be told from a slow binary by the bench line itself"""
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(card, "pp_dpm_sclk")) and not os.path.exists(os.path.join(card, "current_compute_partition")):
            continue
        d = {}
        for key in ("current_compute_partition", "current_memory_partition", "pp_dpm_sclk", "pp_dpm_mclk", "power_dpm_force_performance_level", "gpu_busy_percent"):
            try:
                d[key] = open(os.path.join(card, key)).read().strip().replace("\n", " | ")[:200]
            except OSError:
                pass
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for key in ("power1_cap", "power1_average", "power1_input", "freq1_input", "temp1_input"):
                try:
                    d[key] = int(open(os.path.join(hw, key)).read().strip())
                except (OSError, ValueError):
                    pass
        out[os.path.basename(os.path.dirname(card))] = d
        if len(out) >= 2:
            break
    return out


def budget_check(config, measured):
    """the line's own figures against profiles/budget.json (tools/kernel_budget.py --write: minimum over rounds of a mean launch time
    on the round's reference lease, + 12 %): a lease that runs the same binary slower than that shows up IN the record"""
    try:
        b = json.load(open(os.path.join(ROOT, "profiles", "budget.json")))["allowed_us"]
    except (OSError, KeyError, ValueError):
        return None
    rows = {k: (v, b[f"{config}.{k}"]) for k, v in measured.items() if v is not None and f"{config}.{k}" in b}
    over = {k: {"measured_us": round(v, 3), "allowed_us": a} for k, (v, a) in rows.items() if v > a}
    return {"ok": not over, "over": over, "checked": {k: {"measured_us": round(v, 3), "allowed_us": a} for k, (v, a) in rows.items()},
            "source": "profiles/budget.json"}


def kernel_table(prob, reps):
    """the fused evaluation kernel timed live with HIP events on the library's stream: row 0 is the
    launch one dual evaluation makes (every bucket); the other rows restrict it to one bucket"""
    from cfmm import _lib
    net = prob.net
    parts = []
    for key in ("cp2", "w2", "sum2", "curve2"):
        if key in net and len(net[key]["Ra"]):
            m = len(net[key]["Ra"])
            parts.append((f"eval_kernel[{key} only]", KIND_ID[key], m, m * BYTES_PER_POOL[key]))
    for k, b in sorted(net.get("gn", {}).items()):
        m = b["R"].shape[1]
        if m:
            parts.append((f"eval_kernel[gn{k} only]", -k, m, m * (20 + 20 * k)))
    # (row 0: the fastest of five rounds of `reps` back-to-back launches, as tools/kernel_budget.py takes it -- at C2 the launch is 5.8 us
    #  and the HOST's enqueue rate, 5-8 us per launch and jittery, is what a single round measures: 6.5 and 8.1 us on two runs of one box)
    rows = [dict(kernel="eval_kernel", pools=sum(p[2] for p in parts), bytes=sum(p[3] for p in parts),
                 seconds=min(prob.ctx.time_eval_kernel(_lib.TIME_ALL, reps) for _ in range(5)))]
    if len(parts) > 1:
        for name, code, m, nbytes in parts:
            rows.append(dict(kernel=name, pools=m, bytes=nbytes, seconds=prob.ctx.time_eval_kernel(code, reps)))
    for r in rows:
        r["GBps"] = r["bytes"] / r["seconds"] / 1e9
        r["pools_per_s"] = r["pools"] / r["seconds"]
        r["us"] = r["seconds"] * 1e6
    return rows


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside any launcher: become the launcher.  One rank per GPU of this node,
    rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=os.environ.get("BENCH_CONFIG", "C3"), choices=["C2", "C3", "C4", "C4x4", "C5"])
    ap.add_argument("--zipf", type=float, default=None, help="SURVEY 8(d)'s stress variant: token pairs drawn Zipf(s) hub-weighted instead of uniform (s = 1.1)")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-numpy", action="store_true", help="skip the one-thread NumPy evaluation beside the C baseline")
    ap.add_argument("--no-batch", action="store_true", help="skip the batched-solve figure")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip the pass behind the timed region that re-runs the solves with the shader-clock probe beside them")
    ap.add_argument("--cpu-solves", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--kernel-reps", type=int, default=50)
    ap.add_argument("--allreduce", default=os.environ.get("CFMM_ALLREDUCE", "auto"), choices=["auto", "rccl", "oneshot"],
                    help="the per-evaluation all-reduce: RCCL (default) or the one-shot xGMI exchange (csrc/oneshot.hpp)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="with --gpus N on a one-GPU box: N processes on device 0 (gloo + the one-shot exchange, no RCCL): exercises the whole multi-rank path")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the pool-sharded code path (process group, RCCL communicator, all-reduce per evaluation) even with one rank")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        self_launch(args)                        # (does not return)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if launched and args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks; using {world}", file=sys.stderr)
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
        if args.share_gpu:
            # the ranks are processes on ONE device: gloo carries the host side, the library's per-evaluation exchange goes
            # through the hipIpc-mapped one-shot mailboxes (RCCL refuses two ranks on one device)
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            if torch.cuda.device_count() < world:
                raise SystemExit(f"bench.py: rank {rank} of {world}: --gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible "
                                 "(--share-gpu runs the ranks as processes on one device)")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    shard_kw = dict(dist=dist, device=local_rank, allreduce=("oneshot" if args.share_gpu else args.allreduce))
    if args.share_gpu:
        shard_kw["rccl"] = False

    strong = args.config in ("C4", "C4x4")
    solve_kw = {}
    if strong:
        # strong scaling: ONE fixed network (same seed on every rank), contiguous pool shards
        net = synthetic.config(args.config, seed=0, scale=args.scale, zipf_s=args.zipf)
        total_pools = cfmm.problem.network_pool_count(net)
        utility = cfmm.Arbitrage(net["c"])
        prob = cfmm.distributed.sharded_problem(net, utility, shard=True, **shard_kw)
    else:
        # weak scaling: every rank generates its OWN shard of the config (same tokens / prices / utility)
        net = synthetic.config(args.config, seed=0, scale=args.scale, pool_seed=(rank if world > 1 else None), zipf_s=args.zipf)
        if args.config == "C5":
            # the basket of tools/profile_newton.py: ten tokens worth ~70 units each to be sold for token t (liquidation.py:57,77-80)
            rng = np.random.default_rng(1)
            n = net["n_tokens"]
            h = np.zeros(n); idx = rng.choice(n, 10, replace=False); h[idx] = np.exp(rng.normal(2, 0.5, 10)) / net["prices"][idx] * 10
            t_out = int(rng.integers(0, n)); h[t_out] = 0
            utility = cfmm.Liquidate(h, t_out)
            if args.share_gpu:
                raise SystemExit("bench.py: --config C5 all-reduces a Hessian per Newton step: it needs RCCL, not --share-gpu")
        else:
            utility = cfmm.Arbitrage(net["c"])
        if args.share_gpu:
            solve_kw["method"] = "lbfgs"       # (the second-order fall-back would all-reduce a Hessian: RCCL)
        prob = cfmm.distributed.sharded_problem(net, utility, shard=False, **shard_kw)
        total_pools = prob.m * world
    prob._ensure_ctx()

    def sync():
        if sharded:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    # The per-kernel table of the line (the evaluation launch alone, HIP events on the library's stream, at prices ~1 % off the
    # market: about 88 % of the pools trade, as in the first iterations of a solve and in tools/profile_eval.py, whose rocprofv3
    # trace is committed under profiles/) is measured HERE, in front of the warm-up steps, not behind the timed ones: a few
    # hundred launches that leave the device at its working clock.  With W = 5 cold solves (2.5 ms) alone the first timed steps
    # still ran on a ramping clock -- 20.7 us per iteration where the same binary settles at 20.0 (W = 20).  Nothing of it is
    # inside the timed region, which is K complete cold solves either way.
    prob._send_utility()
    idle_ghz = None
    if rank == 0 and not args.no_clock_probe:                       # the clock of the idle chip, read well in front of everything timed
        try:
            prob.ctx.clock_probe_start(100.0, 50.0)
            time.sleep(0.004)
            idle_ghz = clock_ghz(prob.ctx.clock_probe_stop())
        except Exception as e:
            print(f"bench.py: clock probe unavailable: {e}", file=sys.stderr)
    prob.ctx.set_nu(net["c"] * np.exp(np.random.default_rng(0).normal(0, 0.01, net["n_tokens"])))
    from cfmm import _lib as _l
    t_ramp = time.perf_counter()                          # (the table itself should not be measured on the ramp either: discarded launches,
    while True:                                           #  at least 300 and at least 25 ms of them -- 300 launches of C2's 6 us kernel
        prob.ctx.time_eval_kernel(_l.TIME_ALL, 300)       #  are 2 ms, and its table read 6.5 us where the settled kernel takes 5.8)
        if time.perf_counter() - t_ramp >= 0.025:
            break
    rows = kernel_table(prob, args.kernel_reps)
    for _ in range(args.warmup):
        prob.solve(tol=args.tol, **solve_kw)
    # ... and, single process only, further UNTIMED solves until the host has settled too (reported as `extra_warmup_steps`): the
    # first process on a fresh box has been seen enqueueing launches too slowly to keep the device fed (pageable uploads at 9 GB/s
    # instead of 24, 30.7 us per iteration of device time for a 20.9 us launch) for its first tens of milliseconds.  Steady = the last
    # four solves within 5 % of the fastest seen; bounded by 1 s and 400 solves.  The timed region is still exactly K cold solves.
    # Round 6: the live clock probe showed the criterion above ending on a clock that was still rising (five blocks of four solves:
    # 0.546 ... 0.494 ms per solve at 2.25 ... 2.34 GHz, on a lease where the profiles' 20.1 us launch is reached at the end), and with
    # the clock flat the first block of the timed region still ran 4-8 % behind the other four (0.508-0.519 against 0.468-0.492 ms, with and
    # without the probe: the HOST -- a core that has slept through the kernel table's synchronisations needs tens of milliseconds of
    # back-to-back solves to deliver them at its settled pace).  Now: at least 100 untimed solves, then until the median of the last
    # 20 is no longer 1 % faster than the median of the 20 before them; bounded by 2 s and 3000 solves.
    extra_warmup = 0
    if not sharded:
        seen = []
        t_lim = time.perf_counter() + 2.0
        while extra_warmup < 3000 and time.perf_counter() < t_lim:
            ts = time.perf_counter(); prob.solve(tol=args.tol, **solve_kw); seen.append(time.perf_counter() - ts)
            extra_warmup += 1
            if len(seen) >= 100 and len(seen) % 10 == 0 and float(np.median(seen[-20:])) >= 0.99 * float(np.median(seen[-40:-20])):
                break
    sync()
    evals = 0
    dev_s = 0.0
    newton_steps = 0
    # the K timed steps as (up to) five consecutive blocks: `ms_per_step` = total / K as the contract says; the per-block figures and their
    # median beside it show whether the total is one steady rate or a mean over a drifting one
    nblk = max(1, min(5, args.steps))
    edges = [round(i * args.steps / nblk) for i in range(nblk + 1)]
    block_ms = []
    t0 = time.perf_counter()
    tb = t0
    for b in range(nblk):
        for _ in range(edges[b + 1] - edges[b]):
            prob.solve(tol=args.tol, **solve_kw)
            evals += prob.stats["evals"]
            dev_s += prob.stats["device_seconds"]
            newton_steps += prob.stats.get("newton_steps", 0)
            # the metric is "to 1e-6 rel-gap": both certificates at the requested tolerance, checked here and not only
            # through the status string
            if prob.status != "optimal" or not (prob.gap <= args.tol and prob.infeas <= args.tol):
                raise SystemExit(f"rank {rank}: solve ended with status {prob.status} (gap {prob.gap:.2e}, infeas {prob.infeas:.2e}, tol {args.tol:g})")
        te = time.perf_counter()
        block_ms.append(1e3 * (te - tb) / max(1, edges[b + 1] - edges[b]))
        tb = te
    sync()
    dt = time.perf_counter() - t0
    # The shader-clock probe (include/cfmm.h: cfmm_clock_probe_*): one sleeping wave on a stream of its own that samples {shader cycles,
    # 100 MHz ticks} every 100 us while solves run -- the clock the line's launch durations were measured at.  It runs in a pass of its
    # OWN, the same solves again right behind the timed region, not beside it: the first version sampled during the timed steps, and the
    # second queue cost them 2-5 % (0.500-0.503 ms per solve with the probe against 0.480-0.495 without, same lease, interleaved).  The
    # pass's own ms per solve is reported, so that the cost of looking stays visible.
    live_ghz = None
    chain = None
    clock_pass = None
    if not args.no_clock_probe:                                     # (every rank: the solves of a sharded run are collective; the probe is rank 0's)
        n_pass = int(min(max(args.steps // 2, 20), 100))
        probing = False
        if rank == 0:
            try:
                prob.ctx.clock_probe_start(100.0, 600.0)
                probing = True
            except Exception as e:                                  # (a measurement aid: its absence must not cost the line)
                print(f"bench.py: clock probe unavailable: {e}", file=sys.stderr)
        tp = time.perf_counter()
        for _ in range(n_pass):
            prob.solve(tol=args.tol, **solve_kw)
        tp = time.perf_counter() - tp
        if probing:
            samples = prob.ctx.clock_probe_stop()
            if os.environ.get("CFMM_BENCH_PROBE_DUMP"):                # (diagnostics: the raw {shader cycles, 100 MHz ticks} rows)
                np.save(os.environ["CFMM_BENCH_PROBE_DUMP"], np.asarray(samples))
            live_ghz = clock_ghz(samples)
            c_cyc, c_tick, c_n = prob.ctx.clock_probe_chain()
            chain = {"links": c_n, "shader_cycles": c_cyc, "cycles_per_dependent_v_fma_f64": c_cyc / max(c_n, 1), "ghz": (c_cyc / (c_tick * 10.0)) if c_tick > 0 else None}
            clock_pass = {"solves": n_pass, "ms_per_step": 1e3 * tp / n_pass,
                          "note": "the timed steps again, untimed for `value`, with the clock probe beside them: effective_clock_ghz_live is this pass's clock"}
    if sharded:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    subproblems = evals * total_pools          # every rank runs the same number of evaluations over its own pools
    value = subproblems / dt
    # Utilities that leave prices open (liquidation, swap: C5) start from the network's log-price potentials (cfmm/problem.py:
    # _potentials -- a least-squares fit over the pools' marginal prices, a property of the pools alone, solved ONCE per network)
    # shifted to the prices the utility names: O(n) per utility, kept with the utility object after its first solve.  The same
    # steps again with that memo dropped before every solve = what a FIRST solve of a new basket costs; and, separately, the
    # one-off per-network fit (part of setting a network up, like the upload and the token-block ordering of its pools).
    cold_start_ms = potentials_ms = None
    if args.config == "C5" and not sharded:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            utility._start_memo = None
            prob.solve(tol=args.tol, **solve_kw)
        cold_start_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        from cfmm import problem as _pm
        prob.net.pop("_potentials", None)
        t0 = time.perf_counter(); _pm._potentials(prob.net); potentials_ms = 1e3 * (time.perf_counter() - t0)

    # per-iteration split (all ranks: the all-reduce timing is a collective)
    fold_s, ar_s = prob.ctx.time_collective(args.kernel_reps) if sharded else (0.0, 0.0)
    # the evaluation launch at prices ~1 % off the market (about 88 % of the pools trade, as in the first iterations
    # of a solve and in tools/profile_eval.py, whose rocprofv3 trace is committed under profiles/)
    second_order = prob.stats.get("method") == 2
    newton_kernels = None
    if second_order and not sharded:
        newton_kernels = prob.ctx.time_newton_kernels(max(prob.stats.get("barrier_mu", 0.0), 1e-12), 5)      # at the solution just found
    out = None
    if rank == 0:
        dom = rows[0]
        mix = ", ".join(f"{k}={len(prob.net[k]['Ra'])}" for k in ("cp2", "w2", "curve2", "sum2") if k in prob.net)
        if "gn" in prob.net:
            mix += ", gn3-8=" + str(sum(b["R"].shape[1] for b in prob.net["gn"].values()))
        prof = profile_record(args.config, "zipf" if args.zipf else "")
        pk = "iter" if prob.stats.get("method") == 1 and prof["iter"]["rocprof_avg_us"] else "eval"
        us_iter = 1e6 * dev_s / max(evals, 1)
        if strong:
            workload = (f"{args.config}: {total_pools} constant-product pools / {net['n_tokens']} tokens split over {world} GPU(s) "
                        f"({prob.m} per GPU: {mix}), linear-utility arbitrage, cold-start solve to gap,infeas <= {args.tol:g}")
        elif args.config == "C5":
            workload = (f"C5: {prob.m} pools per GPU ({mix}) / {net['n_tokens']} tokens, liquidation of a 10-token basket "
                        f"(liquidation.py:57,77-80), cold-start solve to gap,infeas <= {args.tol:g} (second-order path)")
        else:
            workload = (f"{args.config}: {prob.m} pools per GPU ({mix}) / {net['n_tokens']} tokens, "
                        f"linear-utility arbitrage, cold-start solve to gap,infeas <= {args.tol:g}")
        if args.zipf:
            workload += f"; token pairs Zipf({args.zipf:g}) hub-weighted (SURVEY 8(d) stress variant)"
        if args.share_gpu:
            workload += f"; --share-gpu: the {world} ranks are processes on ONE device (functional run of the multi-rank path, not a scaling number)"
        # the binding roofline of the dominant kernel: the algorithmic bytes against the HBM peak, and the cycles its
        # vector ALUs were issuing (PMC, newest profile) against all SIMD-cycles of the launch; `bound` names the larger
        # `frac` prices the ALGORITHMIC bytes (the contract's definition); `hbm_frac` the bytes the launch loads AS STORED -- the same
        # number unless a bucket carries its compact mirror (>= 1e6 two-asset pools: 21 B instead of 32), where the algorithmic
        # figure can pass what the memory system delivers and only `hbm_frac` is a statement about bandwidth
        stored_bytes = prob.ctx.eval_bytes()
        alg_frac = dom["bytes"] / (us_iter * 1e-6) / 1e9 / HBM_PEAK_GBS
        hbm_frac = stored_bytes / (us_iter * 1e-6) / 1e9 / HBM_PEAK_GBS
        ev_stored_frac = stored_bytes / dom["seconds"] / 1e9 / HBM_PEAK_GBS

        def valu_fraction(pr, live_seconds):
            """vector-issue fraction of a profiled launch.  With GRBM_GUI_ACTIVE in the same PMC summary: VALU-issue cycles /
            (SIMDs x shader-clock cycles of the SAME dispatches) -- profile-only, no clock assumed; otherwise (older
            summaries) the old estimate against 2.4 GHz x the live duration, flagged as such."""
            if not pr["valu_busy"]:
                return None, None
            if pr["sq_busy"] and pr["pmc_ns"]:
                # (NOT against the profiled dispatch's own cycles: a counter pass stretches the dispatch -- 21.8 us on one box, 35.6
                #  on another for the same 20.5 us launch -- and the fraction of a stretched launch says nothing about the live one)
                clk = pr["sq_busy"] / 32.0 / (pr["pmc_ns"] * 1e-9)
                return pr["valu_busy"] / (SIMDS * clk * live_seconds), ("4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x effective clock x the LIVE launch duration); clock = "
                                                                         "SQ_BUSY_CYCLES / 32 shader engines / the dispatch's duration under the same PMC pass = %.2f GHz (%s)" % (clk / 1e9, pr["traffic_file"]))
            return pr["valu_busy"] / (SIMDS * CLOCK_HZ * live_seconds), "estimate: 4 x SQ_ACTIVE_INST_VALU (" + str(pr["traffic_file"]) + ") / (1024 SIMDs x 2.4 GHz x the LIVE launch duration)"
        valu_frac, valu_src = valu_fraction(prof[pk], us_iter * 1e-6)
        ev_valu, _ = valu_fraction(prof["eval"], dom["seconds"])
        eff_clock = (prof[pk]["sq_busy"] / 32.0 / prof[pk]["pmc_ns"]) if (prof[pk]["sq_busy"] and prof[pk]["pmc_ns"]) else None
        # ... and with the clock measured LIVE beside the timed solves (the probe above): the issue cycles are a property of the binary
        # (SQ_ACTIVE_INST_VALU of the committed PMC pass), the SIMD-cycles available are 1024 x the live clock x the live launch duration --
        # no profiler clock in it.  `clock_accounts_for` = how much of the live / profiled launch-time ratio the clock ratio explains.
        valu_frac_live = (prof[pk]["valu_busy"] / (SIMDS * live_ghz * 1e9 * us_iter * 1e-6)) if (prof[pk]["valu_busy"] and live_ghz) else None
        launch_cycles_live = us_iter * 1e-6 * live_ghz * 1e9 if live_ghz else None
        rp_us = prof[pk]["rocprof_avg_us"]
        launch_cycles_profiled = rp_us * 1e-6 * eff_clock * 1e9 if (rp_us and eff_clock) else None
        clock_note = None
        if launch_cycles_live and launch_cycles_profiled:
            clock_note = {"launch_shader_cycles_live": launch_cycles_live, "launch_shader_cycles_profiled": launch_cycles_profiled,
                          "time_ratio_live_over_profiled": us_iter / rp_us, "cycle_ratio_live_over_profiled": launch_cycles_live / launch_cycles_profiled,
                          "note": "launch duration x clock on both sides: a cycle ratio near 1 with a time ratio above it = the same binary on a slower clock"}
        out = {
            "metric": METRIC,
            "value": value, "unit": "pool-subproblems/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "extra_warmup_steps": extra_warmup, "ms_per_step": 1e3 * dt / args.steps,
            "ms_per_step_blocks": block_ms, "ms_per_step_median_block": float(np.median(block_ms)), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "pools_per_gpu": prob.m, "pools_total": total_pools, "tokens": net["n_tokens"], "seed": 0,
                       "parallelism": f"pool-sharding x{world}" if world > 1 else "single GPU",
                       "rccl_ranks": prob.stats.get("n_ranks", 1),
                       "allreduce": getattr(prob, "allreduce", args.allreduce) if sharded else None,
                       "allreduce_note": getattr(prob, "allreduce_note", "") if sharded else None},
            "evals_per_solve": evals / args.steps,
            "device_ms_per_step": 1e3 * dev_s / args.steps,
            "us_per_eval": 1e6 * dt / max(evals, 1),
            "per_iteration_us": {"total_device": us_iter, "evaluation": dom["us"], "fold": 1e6 * fold_s, "allreduce": 1e6 * ar_s,
                                 "update_in_launch": us_iter - dom["us"] - 1e6 * (fold_s + ar_s),
                                 "note": "evaluation = the evaluation launch alone (eval_kernel), fold / all-reduce as back-to-back launches, all with "
                                         "HIP events on the library's stream; the remainder of the device time per iteration is the in-launch nu "
                                         "update (and, per solve, the start kernel / first evaluation / idle run-ahead launches, amortised)"},
            "gap": prob.gap, "infeas": prob.infeas, "objective": prob.value,
            # The dominant kernel of the timed region is iter_kernel: ONE launch per outer iteration = the in-launch nu update
            # (latency-bound, L2 traffic only) + the evaluation of every pool (the streaming part).  Its average duration over
            # the timed region is the device time per iteration (HIP events around the outer loop, one launch per iteration,
            # back to back); the algorithmic bytes are those of the evaluation.  `evaluation_only` is the same tile code
            # launched without the update (eval_kernel, what cfmm_eval_dual runs), timed as back-to-back launches.
            "roofline": {"bound": "valu" if (valu_frac is not None and valu_frac > hbm_frac) else "hbm",
                         "kernel": "iter_kernel (nu update + evaluation, one launch per iteration)" if prob.stats.get("method") == 1 else dom["kernel"],
                         "achieved": dom["bytes"] / (us_iter * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": alg_frac, "hbm_frac": hbm_frac, "valu_frac": valu_frac,
                         "bytes_as_stored_per_launch": stored_bytes,
                         "frac_note": "frac = algorithmic bytes (SURVEY 8(d): 32 B per constant-product pool, ...) / launch duration / 8 TB/s; hbm_frac = the bytes "
                                      "the launch loads as stored (compact mirror of ids and fee where built) / the same -- equal unless a mirror exists",
                         "valu_frac_note": valu_src, "effective_clock_ghz_under_profiler": eff_clock,
                         "effective_clock_ghz_live": live_ghz, "clock_pass": clock_pass, "clock_ghz_idle": idle_ghz,
                         "valu_frac_live_clock": valu_frac_live, "clock_check": clock_note, "clock_probe_fma_chain": chain,
                         "effective_clock_note": "live = shader cycles / wall time sampled every 100 us by one sleeping wave beside the solves of `clock_pass`, right behind the timed region (cfmm_clock_probe_*); "
                                                 "valu_frac_live_clock = 4 x SQ_ACTIVE_INST_VALU (committed PMC pass) / (1024 SIMDs x live clock x live launch duration)",
                         "l2_hit_rate": prof[pk]["l2_hit_rate"],
                         "traffic": prof[pk]["traffic"],
                         "traffic_source": prof[pk]["traffic_file"],
                         "algorithmic_bytes_per_launch": dom["bytes"], "avg_launch_us": us_iter,
                         "rocprof_avg_launch_us": prof[pk]["rocprof_avg_us"], "rocprof_source": prof[pk]["rocprof_file"],
                         "evaluation_only": {"kernel": "eval_kernel", "avg_launch_us": dom["seconds"] * 1e6, "achieved": dom["GBps"],
                                             "frac": dom["GBps"] / HBM_PEAK_GBS, "hbm_frac": ev_stored_frac, "valu_frac": ev_valu,
                                             "bound": "valu" if (ev_valu is not None and ev_valu > ev_stored_frac) else "hbm",
                                             "traffic": prof["eval"]["traffic"],
                                             "rocprof_avg_launch_us": prof["eval"]["rocprof_avg_us"], "rocprof_source": prof["eval"]["rocprof_file"]},
                         "note": TRAFFIC_NOTE[args.config],
                         "all_kernels": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]},
        }
        if second_order:
            # the second-order path: `value` counts its dual evaluations (smoothed and exact) like any other; the time goes
            # into the dense solve of each Newton step, so THAT is the dominant kernel group and its ceiling is the fp64
            # vector rate (no MFMA in it).  The smoothed evaluation is priced against both ceilings beside it.
            out["newton_steps_per_solve"] = newton_steps / args.steps
            if cold_start_ms is not None:
                out["start_prices"] = {"ms_per_step_memoised": 1e3 * dt / args.steps, "ms_per_step_recomputed": cold_start_ms,
                                       "network_potentials_ms_once_per_network": potentials_ms,
                                       "note": "`ms_per_step` / `value` are measured behind the warm-up, with this utility's start prices memoised "
                                               "(cfmm/problem.py: start_prices); `ms_per_step_recomputed` repeats the timed steps with the memo dropped "
                                               "before every solve (a new basket on the same pools: O(n) from the network's potentials); the least-squares "
                                               "fit of those potentials is a one-off per network, timed separately"}
            if newton_kernels:
                nk = newton_kernels
                nr = (net["n_tokens"] + 31) // 32 * 32
                # n^3 / 3 for the factorisation + 2 n^3 / 3 for the inverse factor that rides its launches (chol.hpp, round 4: two
                # 32^3 products per tile, ~n^3 / (6 x 32^3) tiles) -- the price of a back substitution that is one matrix-vector product
                inverse_factor = os.environ.get("CFMM_BACKSUB", "") != "classic"
                useful_flops = nr ** 3 / 3.0                       # the factorisation proper: what a Newton step NEEDS
                flops = useful_flops * (3.0 if inverse_factor else 1.0)
                # which kernel, how many launches: as cfmm_hip.hip: launch_factor chooses them (chol2.hpp takes the block columns in
                # pairs -- CH_NB = 32, rows rounded up to a pair of blocks -- plus one launch that only finishes the inverse factor)
                pairs = os.environ.get("CFMM_CHOL", "") != "single"
                nr2 = (net["n_tokens"] + 63) // 64 * 64
                nlaunch = (nr2 // 64 + (1 if inverse_factor and nr2 // 32 >= 2 else 0)) if pairs else nr2 // 32
                kname = ("chol_step2_kernel (csrc/chol2.hpp: two block columns per launch, fp64 MFMA side products" if pairs else
                         "chol_step_kernel (csrc/chol.hpp: one block column per launch")
                sm_bytes = dom["bytes"] + 16 * sum(len(prob.net[k]["Ra"]) for k in ("cp2", "w2", "curve2") if k in prob.net)     # + the warm starts: 8 B per direction
                tf = flops / nk["factor"] / 1e12
                rp_us, _, rp_file = newton_profile_record("chol_step2_kernel" if pairs else "chol_step_kernel")
                out["roofline"].update({
                    "bound": "valu", "kernel": kname + f"; the dense Cholesky of one Newton step with its inverse factor: {nlaunch} dependent launches)",
                    "launches_per_factorisation": nlaunch,
                    "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VECTOR_PEAK_TFLOPS,
                    "flop_frac": tf / FP64_VECTOR_PEAK_TFLOPS, "useful_flop_frac": useful_flops / nk["factor"] / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                    "valu_frac": None, "hbm_frac": None, "traffic": None, "traffic_source": None,
                    "flop_frac_note": f"flop_frac (= frac): the flops the launches EXECUTE -- n^3/3 of the {nr} x {nr} factorisation + 2n^3/3 of the inverse factor riding along "
                                      "(work added to turn the back substitution into one matrix-vector product) -- / their measured time, against the fp64 vector peak; "
                                      "useful_flop_frac: the n^3/3 a Newton step needs alone / the same time (a flop rate, not the PMC issue fraction the other configs report as valu_frac)",
                    "valu_frac_note": None,
                    "avg_launch_us": nk["factor"] * 1e6, "algorithmic_bytes_per_launch": None,
                    "avg_launch_us_note": "the whole factorisation (the chain of dependent launches), HIP events on the library's stream; "
                                          "rocprof_avg_launch_us = the kernel-trace average of ONE launch of the chain x launches_per_factorisation",
                    "rocprof_avg_per_launch_us": rp_us, "rocprof_avg_launch_us": (rp_us * nlaunch if rp_us else None), "rocprof_source": rp_file,
                    "newton_step_us": {"smoothed_evaluation_with_hessian": nk["smooth_hess"] * 1e6, "smoothed_evaluation": nk["smooth"] * 1e6,
                                       "factorisation": nk["factor"] * 1e6, "back_substitution": nk["backsolve"] * 1e6},
                    "smoothed_evaluation": {"kernel": "smooth_kernel<false>", "avg_launch_us": nk["smooth"] * 1e6,
                                            "algorithmic_bytes_per_launch": sm_bytes, "hbm_frac": sm_bytes / nk["smooth"] / 1e9 / HBM_PEAK_GBS},
                })
        if not sharded and not strong and not args.no_batch and not second_order and "curve2" not in prob.net and "sum2" not in prob.net:
            # B price vectors per pool read (cfmm_solve_batch; the parametric-sweep use of two-asset.py:34-100): B solves
            # of the same pools under B utilities in lock-step.  Not `value` (the metric is quoted on ONE solve of this
            # config): an extra figure in the same unit.
            B = prob.ctx.batch_capacity()
            rng = np.random.default_rng(1)
            us = [cfmm.Arbitrage(net["c"] * np.exp(rng.normal(0, 0.01, net["n_tokens"]))) for _ in range(B)]
            prob.solve_many(us, tol=args.tol, batch=B)
            t0 = time.perf_counter(); bev = 0; bit = 0; bdev = 0.0
            for _ in range(args.steps):
                res = prob.solve_many(us, tol=args.tol, batch=B)
                if any(r["status"] != "optimal" for r in res):
                    raise SystemExit("batched solve: " + ", ".join(r["status"] for r in res))
                bev += sum(r["stats"]["evals"] for r in res); bit += max(r["stats"]["evals"] for r in res)
                bdev += res[0]["stats"]["device_seconds"]
            bdt = time.perf_counter() - t0
            out["batched"] = {"solves_per_batch": B, "value": bev * prob.m / bdt, "unit": "pool-subproblems/s",
                              "ms_per_batch": 1e3 * bdt / args.steps, "ms_per_solve": 1e3 * bdt / args.steps / B,
                              "device_us_per_lockstep_iteration": 1e6 * bdev / max(bit, 1),
                              "device_us_per_solve_iteration": 1e6 * bdev / max(bev, 1),
                              "note": f"{B} cold solves of the same pools under {B} market-value vectors (1 % apart) in lock-step: every "
                                      "outer iteration reads every pool column once and solves each pool at all price vectors "
                                      "(eval_batch_kernel + one update workgroup per solve); wall time includes the host-side start "
                                      "prices and certificate checks of every solve"}
            prob.set_utility(cfmm.Arbitrage(net["c"]))
        # the line's own kernel figures against the committed budget (profiles/budget.json), and what the driver exposes about the device
        bkey = "C4" if args.config == "C4" and world == 1 else args.config
        if not sharded and args.scale == 1.0 and not args.zipf:
            measured = {"eval_kernel_us": dom["us"]}
            if second_order:
                if newton_kernels:
                    measured.update({"smooth_hess_us": newton_kernels["smooth_hess"] * 1e6, "smooth_us": newton_kernels["smooth"] * 1e6,
                                     "factor_plus_backsolve_us": (newton_kernels["factor"] + newton_kernels["backsolve"]) * 1e6})
            else:
                measured["iter_kernel_us_per_iteration"] = us_iter
                if "batched" in out:
                    measured[f"batch{out['batched']['solves_per_batch']}_us_per_lockstep_iteration"] = out["batched"]["device_us_per_lockstep_iteration"]
            out["budget"] = budget_check(bkey, measured)
            out["budget_ok"] = out["budget"]["ok"] if out["budget"] else None
        out["device_state"] = device_state()
        if not sharded and not strong and not second_order:
            # the same solve with the host-buffer hand-over inside the clock (never `value`): a fresh context, the pool columns
            # uploaded from pageable NumPy buffers, utility, one cold solve, prices and psi read back -- through the raw
            # C-ABI calls INTEGRATION.md's stub makes
            from cfmm.problem import KIND2
            u = cfmm.Arbitrage(net["c"])
            rows = []
            for _ in range(5):
                t0 = time.perf_counter()
                c2 = _lib.Context(net["n_tokens"], 0)
                for key, kind in KIND2.items():
                    if key in net:
                        b = net[key]
                        c2.upload_pools2(kind, b["Ra"], b["Rb"], b["fee"], b["ia"], b["ib"], b.get("wa") if key == "w2" else b.get("alpha"))
                for k, b in net.get("gn", {}).items():
                    c2.upload_poolsN(b["idx"], b["R"], b["w"], b["fee"])
                t1 = time.perf_counter()
                c2.set_utility(u.c, u.h, u.ctype); st2 = c2.solve(net["c"], tol=args.tol); c2.get_solution()
                t2 = time.perf_counter()
                rows.append((t2 - t0, t1 - t0, st2["evals"]))
                c2.close()
            rows.sort()
            tot, up, ev2 = rows[len(rows) // 2]
            out["pcie_inclusive"] = {"value": ev2 * prob.m / tot, "unit": "pool-subproblems/s", "ms_total": 1e3 * tot, "ms_create_and_upload": 1e3 * up,
                                     "upload_GBps": dom["bytes"] / up / 1e9, "evals": ev2,
                                     "note": "median of 5: cfmm_create + cfmm_upload_* of the pool columns from pageable host buffers + "
                                             "cfmm_set_utility + one cold cfmm_solve + cfmm_get_solution"}
        if not sharded and not args.no_cpu:
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
            if second_order:
                # the CPU twin of this path is the first-order iteration (the oracle has no second-order method): dual
                # evaluations of the same network at the same prices are what is comparable, timed for the same duration
                t0 = time.perf_counter(); ce = 0
                while time.perf_counter() - t0 < args.cpu_seconds:
                    o.eval(net["prices"]); ce += 1
                cdt = time.perf_counter() - t0
                out["cpu_baseline"] = {"value": ce * prob.m / cdt, "unit": "pool-subproblems/s", "cores": cores, "kind": "port",
                                       "measures": "dual EVALUATIONS only -- not solves to 1e-6 (the only CPU solve of this instance is the "
                                                   "half-hour NumPy barrier-Newton run behind tests/golden/c5_liquidation.json)",
                                       "sample": f"{ce} exact dual evaluations of the same {prob.m}-pool network by oracle/cfmm_oracle.c (OpenMP, {cores} "
                                                 f"threads), {cdt:.1f} s: the oracle has no second-order method, and its first-order iteration needs "
                                                 "thousands of evaluations on this instance (DESIGN.md); cvxpy is not installed in this image",
                                       "single_evaluation_ms": eval_s * 1e3}
                print(json.dumps(out))
                if sharded:
                    dist.barrier(); dist.destroy_process_group()
                return
            o.solve(net["c"], tol=args.tol)          # warm the OpenMP pool
            t0 = time.perf_counter(); ce = 0; ns = 0
            while ns < args.cpu_solves or time.perf_counter() - t0 < args.cpu_seconds:     # >= 10 s: past any cgroup burst allowance
                r = o.solve(net["c"], tol=args.tol); ce += r["evals"]; ns += 1
                if time.perf_counter() - t0 > 3 * args.cpu_seconds:
                    break
            cdt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": ce * prob.m / cdt, "unit": "pool-subproblems/s", "cores": cores, "kind": "port",
                                   "sample": f"{ns} full solves of the same {prob.m}-pool instance to the same tolerance by "
                                             f"oracle/cfmm_oracle.c (OpenMP, {cores} threads: the fastest of the thread counts tried on "
                                             f"the {avail} CPUs this process may use), {cdt:.1f} s; cvxpy (the reference's solver stack) "
                                             "is not installed in this image",
                                   "evals_per_solve": ce / ns, "objective": r["primal_value"],
                                   "single_evaluation_ms": eval_s * 1e3}
            if not getattr(args, "no_numpy", False) and not set(net) & {"curve2", "pow2", "gk", "sum2"} and prob.m <= 20_000_000:      # (~25 temporaries of 8 B per pool)
                out["cpu_baseline"]["numpy_one_thread"] = numpy_one_thread(net)      # (baseline B of BASELINE.md section 4, beside baseline A above)
        print(json.dumps(out))
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
