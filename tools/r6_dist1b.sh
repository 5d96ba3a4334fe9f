#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
  for ipg in 4 3; do
    CFMM_ITERS_PER_GRAPH=$ipg timeout 600 python bench.py --force-dist --no-cpu --no-batch --steps 20 --warmup 5 > $O/dist1_ipg$ipg.json 2> $O/dist1_ipg$ipg.err; echo "rc=$?"
    python - $O/dist1_ipg$ipg.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], 'ms_per_step %.4f' % d['ms_per_step'], 'evals', d['evals_per_solve'], {k: round(v, 2) for k, v in d['per_iteration_us'].items() if k != 'note'})
PY
  done
done
timeout 600 python -m pytest tests -m gpu -q -x -k "shard or dist or allreduce or all_reduce" 2>&1 | tail -3
