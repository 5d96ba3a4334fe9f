#!/bin/bash
# phase / per-workgroup timers of the iteration kernel (tuning build)
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r3b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
V=$R/cfmm-routing-code_amd/cfmm/variants
for cfg in ${CONFIGS:-C3 C4shard}; do
  CFMM_LIB=$V/libcfmm_hip_timers.so timeout 300 python tools/microbench.py --config $cfg --tag timers --solves 4 2>> $O/mb.err | tee -a $O/mb.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print({k: d[k] for k in ('tag', 'config', 'evals', 'dev_us_per_eval', 'eval_all_us') if k in d})
    for k in ('iter_phases(cyc,us)', 'iter_blocks', 'eval_phases(cyc,us)', 'eval_blocks', 'tile_us(avg,max,count)'):
        if k in d: print('   ', k, d[k])
"
done
tail -3 $O/mb.err
