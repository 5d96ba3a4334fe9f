#!/bin/bash
cd "$(dirname "$0")/.."
for ug in 1 8 64 256; do
echo -n "upd_grid=$ug "; CFMM_UPD_GRID=$ug python tools/microbench.py --config C3 --solves 10 | python -c "
import sys, json; r = json.loads(sys.stdin.read()); print(r['status'], 'evals', r['evals'], 'solve_wall_ms %.3f dev_us/eval %.1f eval %.1f' % (r['solve_wall_ms'], r['dev_us_per_eval'], r['eval_all_us']))"
done
