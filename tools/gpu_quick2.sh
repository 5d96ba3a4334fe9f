#!/bin/bash
# quick GPU check: a selection of the parity tests, then the per-iteration cost of a few configs
# (default build, variant builds named in $VARIANTS, and the two-launch iteration)
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/q2; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "${TESTS:-solve or update or shipped or sharded or reproducible}" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-300
: > $O/sweep.jsonl
run() { tag=$1; shift; env "$@" timeout 300 python tools/microbench.py --config $CFG --tag $tag >> $O/sweep.jsonl 2>> $O/sweep.err; }
for CFG in ${CONFIGS:-C3 C4shard}; do
  run default X=1
  for v in $VARIANTS; do run $v CFMM_LIB=$R/cfmm-routing-code_amd/cfmm/variants/libcfmm_hip_$v.so; done
  run unfused CFMM_FUSED=0
done
python - <<'PY'
import json
for l in open('gpurun_out/q2/sweep.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['config'], r['status'], 'evals', r['evals'], 'dev_us/eval %.2f wall_us/eval %.2f eval_all_us %.2f solve_ms %.3f value %.9g gap %.1e infeas %.1e' % (r['dev_us_per_eval'], r['wall_us_per_eval'], r['eval_all_us'], r['solve_wall_ms'], r['value'], r['gap'], r['infeas']))
PY
tail -3 $O/sweep.err
