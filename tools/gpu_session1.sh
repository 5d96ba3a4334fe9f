#!/bin/bash
# GPU session: parity tests, variant sweep, kernel trace, bench.  Everything lands in gpurun_out/.
cd "$(dirname "$0")/.."
O=gpurun_out/s1; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests" ; timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
V=cfmm-routing-code_amd/cfmm/variants
: > $O/sweep.jsonl
for lib in t512w2 t512w4 t256w2 t256w3 t256w4; do
  for sl in 16; do
    CFMM_LIB=$PWD/$V/libcfmm_hip_$lib.so CFMM_SLICES=$sl timeout 300 python tools/microbench.py --tag $lib --buckets >> $O/sweep.jsonl 2>> $O/sweep.err
  done
done
for sl in 4 8 32 64; do
  CFMM_LIB=$PWD/$V/libcfmm_hip_t512w2.so CFMM_SLICES=$sl timeout 300 python tools/microbench.py --tag slices >> $O/sweep.jsonl 2>> $O/sweep.err
done
CFMM_LIB=$PWD/$V/libcfmm_hip_t512w2.so CFMM_UPDATE_GENERIC=1 timeout 300 python tools/microbench.py --tag updgeneric >> $O/sweep.jsonl 2>> $O/sweep.err
CFMM_LIB=$PWD/$V/libcfmm_hip_t512w2.so CFMM_EVAL_GRID_MULT=2 timeout 300 python tools/microbench.py --tag grid2 >> $O/sweep.jsonl 2>> $O/sweep.err
for cfg in C2 C4shard C5; do
  timeout 300 python tools/microbench.py --config $cfg --tag $cfg --buckets >> $O/sweep.jsonl 2>> $O/sweep.err
done
cat $O/sweep.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['lib'], r['env'], r['config'], 'evals', r['evals'], 'dev_us/eval %.1f wall_us/eval %.1f eval_all_us %.2f' % (r['dev_us_per_eval'], r['wall_us_per_eval'], r['eval_all_us']), r.get('buckets', ''))
"
echo "== kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/trace -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu > $OLDPWD/$O/trace_bench.log 2>&1); echo "rocprof rc=$?"
find $O/trace -name "*stats*" | head; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
echo "== bench"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
