#!/bin/bash
# round 2: fused iteration kernel (default: E = 2 variables per thread) -- parity tests, then A/B sweeps:
# E = 1 build, accumulator slices, host run-ahead depth, graph replay instead of eager launches, the two-launch iteration
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r2c; rm -rf $O; mkdir -p $O
V=$R/cfmm-routing-code_amd/cfmm/variants
export TMPDIR=/tmp
echo "== gpu tests (fused)"; timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log | cut -c1-300
: > $O/sweep.jsonl
run() { tag=$1; shift; env "$@" timeout 300 python tools/microbench.py --config $CFG --tag $tag >> $O/sweep.jsonl 2>> $O/sweep.err; }
for CFG in C3 C4shard; do
  run default X=1
  run e1 CFMM_LIB=$V/libcfmm_hip_e1.so
  run s2 CFMM_SLICES=2
  run s8 CFMM_SLICES=8
  run ra2 CFMM_RUN_AHEAD=2
  run ra5 CFMM_RUN_AHEAD=5
  run graph CFMM_FUSED_GRAPH=1
  run unfused CFMM_FUSED=0
done
python - <<'PY'
import json
for l in open('gpurun_out/r2c/sweep.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['config'], r['status'], 'evals', r['evals'], 'dev_us/eval %.2f wall_us/eval %.2f eval_all_us %.2f solve_ms %.3f value %.9g gap %.1e infeas %.1e' % (r['dev_us_per_eval'], r['wall_us_per_eval'], r['eval_all_us'], r['solve_wall_ms'], r['value'], r['gap'], r['infeas']))
PY
tail -5 $O/sweep.err
