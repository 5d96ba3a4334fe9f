#!/bin/bash
# usage: gpu_quick.sh [tests] [timers] [variants...]   -- quick GPU check used during kernel tuning
cd "$(dirname "$0")/.."
O=gpurun_out/q; mkdir -p $O
V=cfmm-routing-code_amd/cfmm/variants
if [ "$1" = "tests" ]; then shift; timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log; fi
: > $O/sweep.jsonl
for cfg in ${CONFIGS:-C3 C2 C4shard}; do
  for lib in "$@"; do
    L=$PWD/$V/libcfmm_hip_$lib.so; [ "$lib" = default ] && L=
    for uv in ${UVARS:-0}; do CFMM_UPDATE_VARIANT=$uv CFMM_LIB=$L timeout 300 python tools/microbench.py --config $cfg --tag $lib-u$uv $MB_ARGS >> $O/sweep.jsonl 2>> $O/sweep.err; done
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/q/sweep.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['config'], r['status'], 'evals', r['evals'], 'dev_us/eval %.1f eval_all_us %.2f solve_ms %.3f' % (r['dev_us_per_eval'], r['eval_all_us'], r['solve_wall_ms']), r.get('buckets', ''))
    for k in ('eval_blocks', 'eval_phases(cyc,us)', 'upd_phases(cyc,us)', 'upd_A_detail(cyc,us)', 'tile_us(avg,max,count)', 'wave_busy_us(min,mean,max)', 'between_tiles_us(mean,max)'):
        if k in r: print('    ', k, r[k])
PY
tail -3 $O/sweep.err
