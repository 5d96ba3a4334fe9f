#!/bin/bash
# kernel trace of two config-5 solves (second-order path): per-kernel averages
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/c5; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/tools/profile_newton.py --solves 2 > $O/trace.log 2>&1; echo "rc=$?"; tail -1 $O/trace.log | cut -c1-400
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls=%-6s avg_us=%8.2f total_ms=%8.3f pct=%s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
