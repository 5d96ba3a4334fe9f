#!/bin/bash
cd "$(dirname "$0")/.."
for ipg in 1 2 3 4 6 8; do
echo -n "ipg=$ipg "; CFMM_ITERS_PER_GRAPH=$ipg python tools/microbench.py --config C3 --solves 10 | python -c "
import sys, json; r = json.loads(sys.stdin.read()); print('evals', r['evals'], 'solve_wall_ms %.3f dev_us/eval %.1f wall_us/eval %.1f' % (r['solve_wall_ms'], r['dev_us_per_eval'], r['wall_us_per_eval']))"
done
