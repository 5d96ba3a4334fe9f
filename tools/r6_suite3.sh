#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r6s
for i in 1 2 3; do python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r6s/suite$i.log; tail -3 gpurun_out/r6s/suite$i.log; done
