#!/bin/bash
# round 6, first GPU contact: the suite, the bench line with the live clock, E = 1 against the tree, phase stamps
cd "$(dirname "$0")/.."
O=gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_C3.json 2> $O/bench_C3.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r6a/bench_C3.json') if l.startswith('{')][-1])
r = d['roofline']
print('ms_per_step', d['ms_per_step'], 'blocks', d['ms_per_step_blocks'], 'median', d['ms_per_step_median_block'], 'extra_warmup', d['extra_warmup_steps'])
print('us/iter', d['per_iteration_us']['total_device'], 'eval', d['per_iteration_us']['evaluation'], 'live GHz', r['effective_clock_ghz_live'], r['effective_clock_ghz_live_blocks'], 'idle', r['clock_ghz_idle'])
print('chain', r['clock_probe_fma_chain'], 'clock_check', r['clock_check'])
print('valu_frac', r['valu_frac'], 'live', r['valu_frac_live_clock'], 'budget', d.get('budget'))
print('device_state', d.get('device_state'))
print('batched', d.get('batched'))
PY
MB_ARGS="--buckets" bash tools/r5_ab.sh "C3 C4shard C2" e1 2>&1 | tail -30
cp gpurun_out/r5ab/ab.jsonl $O/ab_e1.jsonl
V=$PWD/cfmm-routing-code_amd/cfmm/variants
for cfg in C3 C4shard; do CFMM_LIB=$V/libcfmm_hip_timers.so timeout 300 python tools/microbench.py --config $cfg --tag timers >> $O/timers.jsonl 2>> $O/timers.err; done
python - <<'PY'
import json
for l in open('gpurun_out/r6a/timers.jsonl'):
    r = json.loads(l)
    print(r['config'], 'dev_us/eval %.2f' % r['dev_us_per_eval'])
    for k in ('iter_phases(cyc,us)', 'iter_blocks', 'eval_phases(cyc,us)', 'tile_us(avg,max,count)', 'wave_busy_us(min,mean,max)', 'between_tiles_us(mean,max)'):
        if k in r: print('   ', k, r[k])
PY
