#!/bin/bash
# the -m gpu suite (optionally a -k selection), output under gpurun_out/t
cd "$(dirname "$0")/.."
O=gpurun_out/t; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q "$@" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $O/pytest.log | cut -c1-600
