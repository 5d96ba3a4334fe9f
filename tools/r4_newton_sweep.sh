#!/bin/bash
# sweep of the second-order path's tuning knobs on config 5 (prelude length, initial barrier weight, shrink factor)
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python tools/profile_newton.py --solves 6 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['solves']; ms=sorted(x['solve_ms'] for x in s)
print('$*', 'steps', s[-1]['newton_steps'], 'evals', s[-1]['evals'], s[-1]['status'], 'gap %.1e'%s[-1]['gap'], 'median_ms %.3f'%ms[len(ms)//2])"; }
for pre in 0 4 8 12 16; do run CFMM_NEWTON_PRELUDE=$pre; done
for mu in 0.3 0.03 0.01 0.003; do run CFMM_NEWTON_MU0=$mu; done
for pre in 4 8; do for mu in 0.03 0.01; do run CFMM_NEWTON_PRELUDE=$pre CFMM_NEWTON_MU0=$mu; done; done
