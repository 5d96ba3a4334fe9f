#!/bin/bash
# experiment: idle run-ahead launches behind the end of a solve (CFMM_RUN_AHEAD) against ms_per_step
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/ra; export TMPDIR=/tmp
for rep in 1 2 3; do for c in C3 C2; do for ra in 2 3; do
  CFMM_RUN_AHEAD=$ra timeout 600 python bench.py --config $c --no-cpu --no-batch --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', $ra, round(d['ms_per_step'], 4), round(d['per_iteration_us']['total_device'], 2), d['evals_per_solve'])"
done; done; done
