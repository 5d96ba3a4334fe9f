#!/bin/bash
# A/B on ONE box: usage ab.sh "<configs>" <variant>...   (default = the tree's library); optional: TESTLIB=<variant> runs the suite through it first
cd "$(dirname "$0")/.."
O=gpurun_out/ab; mkdir -p $O; : > $O/ab.jsonl
export TMPDIR=/tmp
V=$PWD/cfmm-routing-code_amd/cfmm/variants
if [ -n "$TESTLIB" ]; then CFMM_LIB=$V/libcfmm_hip_$TESTLIB.so timeout 900 python -m pytest tests -m gpu -q -x ${TESTSEL:+-k "$TESTSEL"} > $O/pytest_$TESTLIB.log 2>&1; echo "pytest($TESTLIB) rc=$?"; tail -5 $O/pytest_$TESTLIB.log; fi
CFGS=$1; shift
for rep in 1 2 3; do
  for cfg in $CFGS; do
    for spec in default "$@"; do
      lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=${spec#*:}
      L=$V/libcfmm_hip_$lib.so; [ "$lib" = default ] && L=
      env $envs CFMM_LIB=$L timeout 300 python tools/microbench.py --config $cfg --tag $spec --solves 20 --reps 200 $MB_ARGS >> $O/ab.jsonl 2>> $O/ab.err
    done
  done
done
python - <<'PY'
import json, collections
rows = collections.defaultdict(list)
for l in open('gpurun_out/ab/ab.jsonl'):
    r = json.loads(l)
    rows[(r['config'], r['tag'])].append(r)
for (cfg, tag), rs in rows.items():
    print('%-8s %-28s %s evals %s  dev_us/eval %s  eval_all_us %s  solve_ms %s' % (cfg, tag, rs[0]['status'], rs[0]['evals'], ' '.join('%.2f' % r['dev_us_per_eval'] for r in rs),
          ' '.join('%.2f' % r['eval_all_us'] for r in rs), ' '.join('%.3f' % r['solve_wall_ms'] for r in rs)))
PY
tail -3 $O/ab.err
