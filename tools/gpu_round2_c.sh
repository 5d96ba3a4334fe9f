#!/bin/bash
# round 2: fused iteration kernel -- correctness (tests), phase timers, E = 1 vs 2
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r2c; rm -rf $O; mkdir -p $O
V=$R/cfmm-routing-code_amd/cfmm/variants
export TMPDIR=/tmp
echo "== gpu tests (fused)"; timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log | cut -c1-300
: > $O/sweep.jsonl
for cfg in C3 C4shard; do
  for lib in default e2 timers timers_e2; do
    L=$V/libcfmm_hip_$lib.so; [ "$lib" = default ] && L=
    CFMM_LIB=$L timeout 300 python tools/microbench.py --config $cfg --tag $lib >> $O/sweep.jsonl 2>> $O/sweep.err
  done
  CFMM_FUSED=0 timeout 300 python tools/microbench.py --config $cfg --tag unfused >> $O/sweep.jsonl 2>> $O/sweep.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r2c/sweep.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['config'], r['status'], 'evals', r['evals'], 'dev_us/eval %.2f wall_us/eval %.2f eval_all_us %.2f solve_ms %.3f value %.9g gap %.1e infeas %.1e' % (r['dev_us_per_eval'], r['wall_us_per_eval'], r['eval_all_us'], r['solve_wall_ms'], r['value'], r['gap'], r['infeas']))
    for k in ('iter_phases(cyc,us)', 'eval_phases(cyc,us)'):
        if k in r: print('    ', k, r[k])
PY
tail -5 $O/sweep.err
