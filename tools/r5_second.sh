#!/bin/bash
# round 5: K-asset tile change -- suite, budget, phase timers (the fixed-cost breakdown of an evaluation launch)
cd "$(dirname "$0")/.."
O=gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python tools/kernel_budget.py > $O/budget.log 2>&1; echo "budget rc=$?"; tail -1 $O/budget.log | cut -c1-900
V=$PWD/cfmm-routing-code_amd/cfmm/variants
for cfg in C3 C4shard C2; do
  timeout 300 python tools/microbench.py --config $cfg --tag default --buckets >> $O/mb.jsonl 2>> $O/mb.err
  CFMM_LIB=$V/libcfmm_hip_timers.so timeout 300 python tools/microbench.py --config $cfg --tag timers >> $O/mb.jsonl 2>> $O/mb.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r5b/mb.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['config'], r['status'], 'evals', r['evals'], 'dev_us/eval %.2f eval_all_us %.2f solve_ms %.3f' % (r['dev_us_per_eval'], r['eval_all_us'], r['solve_wall_ms']), r.get('buckets', ''))
    for k in ('eval_blocks', 'eval_phases(cyc,us)', 'iter_phases(cyc,us)', 'iter_blocks', 'tile_us(avg,max,count)', 'wave_busy_us(min,mean,max)', 'between_tiles_us(mean,max)'):
        if k in r: print('    ', k, r[k])
PY
tail -3 $O/mb.err
