#!/usr/bin/env python
"""Copies the judged summaries of the last `tools/gpu_profile.sh` run (gpurun_out/p, scratch) into profiles/
(tracked): bench lines, rocprofv3 --kernel-trace --stats tables, per-kernel PMC medians.
Empty the LOCAL gpurun_out/p before the gpurun call: gpurun merges new files into it and leaves old ones where they are."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "p")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)

for name in ("bench.json", "bench_C2.json", "bench_C3.json", "bench_C4.json", "bench_C5.json", "bench_C4x4.json", "bench_C3zipf.json", "bench_dist1.json", "bench_share2.json", "upload.json",
             "kernel_durations.json", "small.json", "batch_C3.jsonl", "batch_C4shard.jsonl", "host_overhead.json", "table.jsonl", "ulog.json", "table_newton.jsonl"):
    if os.path.exists(os.path.join(SRC, name)):
        if name.startswith("bench"):                 # the JSON line alone (a multi-rank run's stdout also carries gloo's banner)
            lines = [l for l in open(os.path.join(SRC, name)).read().splitlines() if l.startswith("{")]
            open(os.path.join(DST, f"{tag}_{name}"), "w").write("\n".join(lines[-1:]) + "\n")
        else:
            shutil.copy(os.path.join(SRC, name), os.path.join(DST, f"{tag}_{name}"))
for d in sorted(glob.glob(os.path.join(SRC, "trace_*"))):
    name = os.path.basename(d)[len("trace_"):]
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(DST, f"{tag}_rocprofv3_kernel_stats_{name}.csv"))
rows = []
for f in sorted(glob.glob(os.path.join(SRC, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    cfg = os.path.basename(os.path.dirname(f) if os.path.basename(os.path.dirname(f)).startswith("pmc_") else os.path.dirname(os.path.dirname(f)))
    cfg = [p for p in f.split(os.sep) if p.startswith("pmc_")][0].split("_")[1]
    agg = {}
    seen = set()
    for r in csv.DictReader(open(f)):
        agg.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        # the dispatch's duration UNDER THIS PMC PASS, once per dispatch and pass (pseudo-counter DISPATCH_NS@<first counter of the
        # pass>): what a counter of the same pass has to be divided by to become a rate
        if (r["Dispatch_Id"], r["Kernel_Name"]) not in seen and r["Counter_Name"] in ("SQ_BUSY_CYCLES",):
            seen.add((r["Dispatch_Id"], r["Kernel_Name"]))
            agg.setdefault((r["Kernel_Name"], "DISPATCH_NS_of_the_SQ_BUSY_CYCLES_pass"), []).append(float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for (kn, cn), v in sorted(agg.items()):
        v.sort()
        rows.append(dict(config=cfg, kernel=kn, counter=cn, dispatches=len(v), median=v[len(v) // 2], min=v[0], max=v[-1]))
with open(os.path.join(DST, f"{tag}_rocprofv3_pmc_medians.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["config", "kernel", "counter", "dispatches", "median", "min", "max"])
    w.writeheader(); w.writerows(rows)
print("profiles/:", sorted(os.listdir(DST)))
