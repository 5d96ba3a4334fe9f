#!/bin/bash
# A/B of the second-order loop's host <-> device hand-off (handoff.hpp): CFMM_NEWTON_IO=blit is the hipMemcpyAsync / hipMemsetAsync path
cd "$(dirname "$0")/.."
O=gpurun_out/nab; mkdir -p $O
for io in lean blit lean blit; do
  CFMM_NEWTON_IO=$io timeout 300 python tools/profile_newton.py --solves 8 2>$O/err_$io.log | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['solves']
print('$io', 'steps', s[-1]['newton_steps'], 'evals', s[-1]['evals'], 'status', s[-1]['status'], 'gap %.2e'%s[-1]['gap'], 'solve_ms', ' '.join('%.3f'%x['solve_ms'] for x in s), 'host_ms', ' '.join('%.3f'%x['host_ms'] for x in s[-3:]))"
done
if [ "$1" = tests ]; then timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log; fi
