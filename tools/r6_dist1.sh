#!/bin/bash
# the pool-sharded code path with ONE rank (bench.py --force-dist): the fold launch in front of RCCL (rounds 2-5) against the slices all-reduced as they are
cd "$(dirname "$0")/.."
O=gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for fold in 0 1; do
    CFMM_RCCL_FOLD=$fold timeout 600 python bench.py --force-dist --no-cpu --no-batch --steps 20 --warmup 5 > $O/dist1_fold$fold.json 2> $O/dist1_fold$fold.err; echo "rc=$?"
    python - $O/dist1_fold$fold.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], 'ms_per_step %.4f' % d['ms_per_step'], 'evals', d['evals_per_solve'], {k: round(v, 2) for k, v in d['per_iteration_us'].items() if k != 'note'})
PY
  done
done
timeout 600 python -m pytest tests -m gpu -q -x -k "shard or dist or allreduce or all_reduce" 2>&1 | tail -3
