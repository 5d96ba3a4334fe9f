#!/bin/bash
# round-4 A/B probe: parity suite, then the staged (LDS-DMA) tile walk against the direct one, per config
cd "$(dirname "$0")/.."
O=gpurun_out/r4; mkdir -p $O
export TMPDIR=/tmp
if [ "$1" = "tests" ]; then shift; timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log | cut -c1-300; fi
: > $O/sweep.jsonl
V=cfmm-routing-code_amd/cfmm/variants
for cfg in ${CONFIGS:-C3 C4shard C4 C2}; do
  for dma in ${DMAS:-1 0}; do
    for lib in ${LIBS:-default}; do
      L=$PWD/$V/libcfmm_hip_$lib.so; [ "$lib" = default ] && L=
      CFMM_TILE_DMA=$dma CFMM_LIB=$L timeout 300 python tools/microbench.py --config $cfg --tag $lib-dma$dma $MB_ARGS >> $O/sweep.jsonl 2>> $O/sweep.err
    done
  done
done
for cfg in ${TIMER_CONFIGS:-}; do for dma in 1 0; do CFMM_TILE_DMA=$dma CFMM_LIB=$PWD/$V/libcfmm_hip_timers.so timeout 300 python tools/microbench.py --config $cfg --tag timers-dma$dma >> $O/sweep.jsonl 2>> $O/sweep.err; done; done
python - <<'PY'
import json
for l in open('gpurun_out/r4/sweep.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['config'], r['status'], 'evals', r['evals'], 'dev_us/eval %.2f eval_all_us %.2f solve_ms %.3f' % (r['dev_us_per_eval'], r['eval_all_us'], r['solve_wall_ms']))
    for k in ('iter_phases(cyc,us)', 'iter_blocks'):
        if k in r: print('    ', k, r[k])
PY
tail -5 $O/sweep.err
