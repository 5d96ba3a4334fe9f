#!/bin/bash
# round 5, first GPU contact: the whole -m gpu suite, the small-network timings (the swept two-asset.py), the kernel budget
cd "$(dirname "$0")/.."
O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 300 python tools/small_timing.py > $O/small.json 2> $O/small.err; echo "small rc=$?"; cat $O/small.json | cut -c1-1500
timeout 600 python tools/kernel_budget.py > $O/budget.log 2>&1; echo "budget rc=$?"; tail -5 $O/budget.log | cut -c1-800
