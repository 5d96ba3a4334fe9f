#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/timers; mkdir -p $O; : > $O/t.jsonl
export TMPDIR=/tmp
V=$PWD/cfmm-routing-code_amd/cfmm/variants
for rep in 1 2 3; do for lib in "$@"; do CFMM_LIB=$V/libcfmm_hip_$lib.so timeout 300 python tools/microbench.py --config ${CFG:-C3} --tag $lib >> $O/t.jsonl 2>> $O/t.err; done; done
python - <<'PY'
import json
for l in open('gpurun_out/timers/t.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['config'], 'dev_us/eval %.2f eval_all_us %.2f' % (r['dev_us_per_eval'], r['eval_all_us']))
    for k in ('eval_phases(cyc,us)', 'iter_phases(cyc,us)', 'iter_blocks'):
        if k in r: print('    ', k, {a: (b[1] if isinstance(b, list) and len(b) == 2 else b) for a, b in r[k].items()})
PY
