#!/bin/bash
# the same fuzzers on seed ranges the round had not seen: eight processes side by side on one device.
#   fuzz_campaign_extended.sh [K]     K = 0 (default): fuzz_small 2000-2599, fuzz_table 1000-1599, fuzz_mid 120-359  -> profiles/r06_fuzz_extended.txt
#                                     K = 1, 2, ...: the ranges shifted by K x (1000, 1000, 300)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp OPENBLAS_NUM_THREADS=1
K=${1:-0}
S=$((2000 + 1000 * K)); T=$((1000 + 1000 * K)); M=$((120 + 300 * K))
O=gpurun_out/r6g$K; mkdir -p $O
(python tools/fuzz_small.py $S 200 > $O/fuzz_small_a.txt 2>&1 &
 python tools/fuzz_small.py $((S + 200)) 200 > $O/fuzz_small_b.txt 2>&1 &
 python tools/fuzz_small.py $((S + 400)) 200 > $O/fuzz_small_c.txt 2>&1 &
 python tools/fuzz_table.py $T 120 > $O/fuzz_table_a.txt 2>&1 &
 python tools/fuzz_table.py $((T + 120)) 120 > $O/fuzz_table_b.txt 2>&1 &
 python tools/fuzz_table.py $((T + 240)) 120 > $O/fuzz_table_c.txt 2>&1 &
 python tools/fuzz_table.py $((T + 360)) 240 > $O/fuzz_table_d.txt 2>&1 &
 python tools/fuzz_mid.py $M 240 > $O/fuzz_mid.txt 2>&1 &
 wait)
for f in $O/fuzz_*.txt; do echo "## $f"; grep -v "Warning\|^  " $f | tail -6 | cut -c1-330; done
