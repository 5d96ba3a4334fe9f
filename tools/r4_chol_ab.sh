#!/bin/bash
# A/B of the factorisation: two block columns per launch (chol2.hpp) against one (CFMM_CHOL=single)
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_newton.py -q -x 2>&1 | tail -4
for ch in pairs single pairs single; do
  CFMM_CHOL=$ch timeout 300 python tools/profile_newton.py --solves 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['solves']
print('$ch', 'steps', s[-1]['newton_steps'], 'evals', s[-1]['evals'], s[-1]['status'], 'gap %.2e'%s[-1]['gap'], 'value %.9f'%s[-1]['value'], 'solve_ms', ' '.join('%.3f'%x['solve_ms'] for x in s))"
done
for ch in pairs single; do CFMM_CHOL=$ch python tools/kernel_budget.py --only C5 2>&1 | tail -2; done
