#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s3; mkdir -p $O
V=cfmm-routing-code_amd/cfmm/variants
: > $O/sweep.jsonl
for cfg in C3 C2 C4shard; do
CFMM_LIB=$PWD/$V/libcfmm_hip_timers.so timeout 300 python tools/microbench.py --config $cfg --tag timers >> $O/sweep.jsonl 2>> $O/sweep.err
done
cat $O/sweep.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['config'], r['status'], 'evals', r['evals'], 'dev_us/eval %.1f eval_all_us %.2f' % (r['dev_us_per_eval'], r['eval_all_us']))
    print('   eval', r.get('eval_phases(cyc,us)'))
    print('   upd ', r.get('upd_phases(cyc,us)'))
"
tail -5 $O/sweep.err
rocm-smi --showclocks 2>/dev/null | head -20
