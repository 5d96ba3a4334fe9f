#!/bin/bash
# round 2: the fused iteration kernel (one launch per iteration) against the two-launch iteration
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/r2b; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests (fused)"; timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
: > $O/sweep.jsonl
for cfg in ${CONFIGS:-C3 C2 C4shard C4}; do
  for f in 1 0; do
    CFMM_FUSED=$f timeout 300 python tools/microbench.py --config $cfg --tag fused$f >> $O/sweep.jsonl 2>> $O/sweep.err
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r2b/sweep.jsonl'):
    r = json.loads(l)
    print(r['tag'], r['config'], r['status'], 'evals', r['evals'], 'dev_us/eval %.2f wall_us/eval %.2f eval_all_us %.2f solve_ms %.3f value %.9g gap %.1e infeas %.1e' % (r['dev_us_per_eval'], r['wall_us_per_eval'], r['eval_all_us'], r['solve_wall_ms'], r['value'], r['gap'], r['infeas']))
PY
tail -5 $O/sweep.err
echo "== bench C3"; timeout 600 python bench.py --no-cpu > $O/bench_c3.json 2> $O/bench_c3.err; echo "rc=$?"; cut -c1-1400 $O/bench_c3.json; tail -3 $O/bench_c3.err
cd /tmp
for f in 1 0; do
  CFMM_FUSED=$f timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f$f -o t -- python $R/tools/microbench.py --config C3 --solves 20 > $O/trace_f$f.log 2>&1
  echo "== kernel stats fused=$f"; find $O/trace_f$f -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 | head -8
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +3M -delete
