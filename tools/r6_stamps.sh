#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
V=$PWD/cfmm-routing-code_amd/cfmm/variants
for v in st0b st1b st0u st1u; do echo "== $v"; CFMM_LIB=$V/libcfmm_hip_$v.so timeout 200 python tools/ch2_stamps.py 2>&1 | tail -4; done
bash tools/r6_ab.sh "C3 C2" nofred 2>&1 | tail -8
python bench.py 2>&1 | tail -1 > gpurun_out/r6_bench_now.json; python - <<'PY'
import json; r = json.loads(open('gpurun_out/r6_bench_now.json').read())
print({k: r.get(k) for k in ('value', 'ms_per_step', 'ms_per_step_blocks', 'ms_per_step_median_block', 'budget_ok', 'extra_warmup')})
print({k: r['roofline'].get(k) for k in ('frac', 'effective_clock_ghz_live', 'effective_clock_ghz_live_blocks', 'clock_ghz_idle', 'clock_check', 'kernel_avg_us')})
PY
