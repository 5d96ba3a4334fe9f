#!/bin/bash
# round 3, first measurement: parity tests, then A/B of the iteration kernel (r02 build vs. the one-wave scalar section +
# table-free tile prologue, with and without the ping-pong walk) on C3 / C4 shard / C4, phase timers on C3
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r3a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
V=$R/cfmm-routing-code_amd/cfmm/variants
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
mb() { # tag lib config [env...]
  tag=$1; lib=$2; cfg=$3; shift 3
  env CFMM_LIB=$lib "$@" timeout 300 python tools/microbench.py --config $cfg --tag $tag --solves 8 2>> $O/mb.err | tee -a $O/mb.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print({k: d[k] for k in ('tag', 'config', 'evals', 'status', 'dev_us_per_eval', 'wall_us_per_eval', 'eval_all_us', 'solve_wall_ms') if k in d})
    for k in ('iter_phases(cyc,us)', 'iter_blocks', 'eval_phases(cyc,us)', 'eval_blocks'):
        if k in d: print('   ', k, d[k])
"
}
for cfg in ${CONFIGS:-C3 C4shard C4 C2}; do
  mb r02 $V/libcfmm_hip_r02.so $cfg
  mb new $R/cfmm-routing-code_amd/cfmm/libcfmm_hip.so $cfg
  mb new_nopp $R/cfmm-routing-code_amd/cfmm/libcfmm_hip.so $cfg CFMM_PINGPONG=0
done
mb timers $V/libcfmm_hip_timers.so C3
mb timers $V/libcfmm_hip_timers.so C4shard
