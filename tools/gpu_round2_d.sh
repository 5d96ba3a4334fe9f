#!/bin/bash
# round 2: reproducible mode + cvx shim on the GPU; cost of the reproducible mode
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r2d; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log | cut -c1-400
: > $O/sweep.jsonl
run() { tag=$1; shift; env "$@" timeout 300 python tools/microbench.py --config $CFG --tag $tag >> $O/sweep.jsonl 2>> $O/sweep.err; }
for CFG in C3 C4shard; do
  run default X=1
  run det CFMM_DETERMINISTIC=1
done
python - <<'PY'
import json
for l in open('gpurun_out/r2d/sweep.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['config'], r['status'], 'evals', r['evals'], 'dev_us/eval %.2f wall_us/eval %.2f eval_all_us %.2f solve_ms %.3f value %.12g gap %.1e infeas %.1e' % (r['dev_us_per_eval'], r['wall_us_per_eval'], r['eval_all_us'], r['solve_wall_ms'], r['value'], r['gap'], r['infeas']))
PY
tail -5 $O/sweep.err
