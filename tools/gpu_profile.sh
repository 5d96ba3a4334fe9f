#!/bin/bash
# Round artefacts: parity tests, bench lines, rocprofv3 kernel traces + PMC passes.  Results -> gpurun_out/p/
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/p; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests"; timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
echo "== bench"; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-900 $O/bench.json
echo "== bench C4"; timeout 900 python bench.py --config C4 --no-cpu > $O/bench_C4.json 2> $O/bench_C4.err; echo "bench rc=$?"; cut -c1-600 $O/bench_C4.json
echo "== bench (pool-sharded path, one rank)"; timeout 900 python bench.py --force-dist --no-cpu > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?"
echo "== upload"; timeout 300 python tools/upload_timing.py > $O/upload.json 2> $O/upload.err; cat $O/upload.json
echo "== small networks"; timeout 300 python tools/small_timing.py > $O/small.json 2> $O/small.err; cut -c1-600 $O/small.json
echo "== batched solves"; for c in C3 C4shard; do timeout 600 python tools/batch_timing.py --config $c > $O/batch_$c.jsonl 2> $O/batch_$c.err; cut -c1-300 $O/batch_$c.jsonl; done
cd /tmp
echo "== kernel trace of the batched solve (B = 8, C3)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_batch -o t -- python $R/tools/batch_timing.py --config C3 --sizes 8 --reps 3 > $O/trace_batch.log 2>&1; echo "rc=$?"
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_batch_$tag -o c -- python $R/tools/batch_timing.py --config C3 --sizes 8 --reps 2 > $O/pmc_batch_$tag.log 2>&1; echo "pmc batch $tag rc=$?"
done
echo "== kernel trace of the reference-sized solves"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_small -o t -- python $R/tools/small_timing.py > $O/trace_small.log 2>&1; echo "rc=$?"
echo "== kernel trace of bench.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $O/trace_bench.log 2>&1; echo "rc=$?"
PMCS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum")
for cfg in C3 C4; do
  for tgt in eval iter; do
    [ $tgt = eval ] && CMD="python $R/tools/profile_eval.py --config $cfg" || CMD="python $R/tools/profile_iter.py --config $cfg --solves 6"
    name=$cfg; [ $tgt = iter ] && name=${cfg}iter
    echo "== $name kernel trace"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o t -- $CMD > $O/trace_$name.log 2>&1; echo "rc=$?"; tail -1 $O/trace_$name.log | cut -c1-300
    for pmc in "${PMCS[@]}"; do
      tag=$(echo $pmc | cut -d' ' -f1)
      timeout 900 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_${name}_$tag -o c -- $CMD > $O/pmc_${name}_$tag.log 2>&1; echo "pmc $name $tag rc=$?"
    done
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
O = 'gpurun_out/p'
for f in sorted(glob.glob(O + '/trace_*/**/*kernel_stats.csv', recursive=True)):
    print(f); print(open(f).read()[:900])
for f in sorted(glob.glob(O + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'eval_kernel' in r['Kernel_Name'] or 'iter_kernel' in r['Kernel_Name'] or 'eval_batch' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:46], r['Counter_Name'])].append(float(r['Counter_Value']))
    print(f)
    for (kn, cn), v in sorted(agg.items()):
        v = sorted(v)
        print('   %-48s %-22s n=%-4d median=%.5g min=%.5g max=%.5g' % (kn, cn, len(v), v[len(v) // 2], v[0], v[-1]))
PY
# per-dispatch durations of the iteration kernel (median / mean over the full launches) next to the stats tables
python - <<'PY'
import csv, glob, json
out = {}
for f in sorted(glob.glob('gpurun_out/p/trace_*/**/*kernel_trace.csv', recursive=True)):
    name = [p for p in f.split('/') if p.startswith('trace_')][0][6:]
    d = {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'iter_kernel' in k or 'eval_kernel' in k or 'update' in k or 'eval_batch' in k or 'solve_tiny' in k:
            d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out[name] = {}
    for k, v in d.items():
        v.sort()
        full = [x for x in v if x > 0.6 * v[-1]]
        out[name][k] = dict(calls=len(v), median_us=v[len(v) // 2], mean_us=sum(v) / len(v), full_launches=len(full), full_mean_us=sum(full) / len(full), min_us=v[0], max_us=v[-1])
json.dump(out, open('gpurun_out/p/kernel_durations.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
du -sh $O
