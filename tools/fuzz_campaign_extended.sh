#!/bin/bash
# the same fuzzers on seed ranges the round had not seen (small 2000-2599, table 1000-1599, mid 120-359): eight processes side by side -> profiles/r06_fuzz_extended.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp OPENBLAS_NUM_THREADS=1

O=gpurun_out/r6g; mkdir -p $O
(python tools/fuzz_small.py 2000 200 > $O/fuzz_small_a.txt 2>&1 &
 python tools/fuzz_small.py 2200 200 > $O/fuzz_small_b.txt 2>&1 &
 python tools/fuzz_small.py 2400 200 > $O/fuzz_small_c.txt 2>&1 &
 python tools/fuzz_table.py 1000 120 > $O/fuzz_table_a.txt 2>&1 &
 python tools/fuzz_table.py 1120 120 > $O/fuzz_table_b.txt 2>&1 &
 python tools/fuzz_table.py 1240 120 > $O/fuzz_table_c.txt 2>&1 &
 python tools/fuzz_table.py 1360 240 > $O/fuzz_table_d.txt 2>&1 &
 python tools/fuzz_mid.py 120 240 > $O/fuzz_mid.txt 2>&1 &
 wait)
