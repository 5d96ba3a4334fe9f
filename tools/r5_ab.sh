#!/bin/bash
# A/B on ONE box: the working tree's library against variants (interleaved, twice)   usage: r5_ab.sh "C3 C4shard" base [other...]
cd "$(dirname "$0")/.."
O=gpurun_out/r5ab; mkdir -p $O; : > $O/ab.jsonl
export TMPDIR=/tmp
V=$PWD/cfmm-routing-code_amd/cfmm/variants
CFGS=$1; shift
for rep in 1 2 3; do
  for cfg in $CFGS; do
    for lib in default "$@"; do
      L=$V/libcfmm_hip_$lib.so; [ "$lib" = default ] && L=
      CFMM_LIB=$L timeout 300 python tools/microbench.py --config $cfg --tag $lib --solves 20 --reps 200 $MB_ARGS >> $O/ab.jsonl 2>> $O/ab.err
    done
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r5ab/ab.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['config'], r['status'], 'evals', r['evals'], 'dev_us/eval %.2f eval_all_us %.2f solve_ms %.3f' % (r['dev_us_per_eval'], r['eval_all_us'], r['solve_wall_ms']), r.get('buckets', ''))
PY
tail -3 $O/ab.err
