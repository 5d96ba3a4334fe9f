#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s2; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests" ; timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
V=cfmm-routing-code_amd/cfmm/variants
: > $O/sweep.jsonl
for sl in 1 2 4 8; do
  CFMM_SLICES=$sl timeout 300 python tools/microbench.py --tag slices$sl >> $O/sweep.jsonl 2>> $O/sweep.err
done
for lib in t256w2 t512w4; do
  CFMM_LIB=$PWD/$V/libcfmm_hip_$lib.so timeout 300 python tools/microbench.py --tag $lib --buckets >> $O/sweep.jsonl 2>> $O/sweep.err
done
timeout 300 python tools/microbench.py --tag default --buckets >> $O/sweep.jsonl 2>> $O/sweep.err
for cfg in C2 C4shard; do
  timeout 300 python tools/microbench.py --config $cfg --tag $cfg >> $O/sweep.jsonl 2>> $O/sweep.err
done
cat $O/sweep.jsonl | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    print(r['tag'], r['lib'], r['env'], r['config'], r['status'], 'evals', r['evals'], 'dev_us/eval %.1f wall_us/eval %.1f eval_all_us %.2f' % (r['dev_us_per_eval'], r['wall_us_per_eval'], r['eval_all_us']), r.get('buckets', ''))
"
tail -5 $O/sweep.err
echo "== kernel trace"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/trace -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu > $OLDPWD/$O/trace_bench.log 2>&1); echo "rocprof rc=$?"
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('gpurun_out/s2/trace/*.db')[0])
for r in db.execute("select name, count(*), sum(end-start)/1000.0, avg(end-start)/1000.0, min(end-start)/1000.0, max(end-start)/1000.0 from kernels group by name order by 3 desc"):
    print("%-60s %6d %10.1f %8.2f %8.2f %8.2f" % (r[0][:60], r[1], r[2], r[3], r[4], r[5]))
PY
