#!/bin/bash
cd "$(dirname "$0")/.."
for env in "X=1" "OPENBLAS_NUM_THREADS=1" "X=1" "OPENBLAS_NUM_THREADS=1"; do
  env $env timeout 300 python tools/profile_newton.py --solves 40 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=[x['solve_ms'] for x in d['solves']]
print('$env', 'median %.3f'%sorted(s)[len(s)//2], 'outliers', [round(x,1) for x in s if x>8])"
done
