#!/bin/bash
# fuzz campaigns with the referee that answers (oracle/dual_np.py), seven processes side by side on one device; artefacts under gpurun_out/r6f -> profiles/r06_fuzz_*
cd "$(dirname "$0")/.."
export TMPDIR=/tmp OPENBLAS_NUM_THREADS=1
O=gpurun_out/r6f; mkdir -p $O
(python tools/fuzz_small.py 1000 200 > $O/fuzz_small_a.txt 2>&1 &
 python tools/fuzz_small.py 1200 200 > $O/fuzz_small_b.txt 2>&1 &
 python tools/fuzz_small.py 1400 200 > $O/fuzz_small_c.txt 2>&1 &
 python tools/fuzz_table.py 0 120 > $O/fuzz_table_a.txt 2>&1 &
 python tools/fuzz_table.py 334 120 > $O/fuzz_table_b.txt 2>&1 &
 python tools/fuzz_table.py 667 120 > $O/fuzz_table_c.txt 2>&1 &
 python tools/fuzz_mid.py 0 120 > $O/fuzz_mid.txt 2>&1 &
 wait)
head -1 $O/fuzz_*.txt
