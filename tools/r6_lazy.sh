#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
bash tools/r6_ab.sh "C3 C4shard C2" eager 2>&1 | tail -10
