#!/bin/bash
# selected tests + the phase timers of the one-launch iteration
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/q3; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "${TESTS:-batched or one_shot or c4_single}" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-400
for CFG in ${CONFIGS:-C3 C4shard}; do
  CFMM_LIB=$R/cfmm-routing-code_amd/cfmm/variants/libcfmm_hip_timers.so timeout 300 python tools/microbench.py --config $CFG --tag timers 2> $O/mb_$CFG.err | tee $O/mb_$CFG.json | cut -c1-2500
done
