#!/bin/bash
# Round artefacts: parity tests, bench line, rocprofv3 kernel trace + PMC passes.  Results -> gpurun_out/p/
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/p; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
echo "== bench"; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-700 $O/bench.json
cd /tmp
echo "== kernel trace of bench.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $O/trace_bench.log 2>&1; echo "rc=$?"
for cfg in C3 C4; do
  CMD="python $R/tools/profile_eval.py --config $cfg"
  echo "== $cfg kernel trace"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$cfg -o t -- $CMD > $O/trace_$cfg.log 2>&1; echo "rc=$?"; tail -1 $O/trace_$cfg.log
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    timeout 900 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_${cfg}_$tag -o c -- $CMD > $O/pmc_${cfg}_$tag.log 2>&1; echo "pmc $cfg $tag rc=$?"
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
O = 'gpurun_out/p'
for f in sorted(glob.glob(O + '/trace_*/**/*kernel_stats.csv', recursive=True)):
    print(f); print(open(f).read()[:1500])
for f in sorted(glob.glob(O + '/pmc_*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'eval_kernel' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:46], r['Counter_Name'])].append(float(r['Counter_Value']))
    print(f)
    for (kn, cn), v in sorted(agg.items()):
        v = sorted(v)
        print('   %-48s %-22s n=%-4d median=%.5g min=%.5g max=%.5g' % (kn, cn, len(v), v[len(v) // 2], v[0], v[-1]))
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
du -sh $O
