#!/bin/bash
# Round artefacts: parity tests, one bench line per BASELINE config (+ the HBM-streaming C4x4 set and the Zipf(1.1) stress
# variant of C3), rocprofv3 kernel traces + SEPARATE PMC passes (never combined with a trace: MI355X_MICROARCH.md / gpurun).
# Results -> gpurun_out/p/ (scratch); tools/collect_profiles.py <tag> copies the judged summaries into profiles/.
# usage: gpu_profile.sh [quick]     (quick: no pytest, no side timings)
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/p; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
if [ "$1" != quick ]; then
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
fi
for c in C3 C2 C4 C5 C4x4; do
  echo "== bench $c"; timeout 1200 python bench.py --config $c --cpu-seconds 8 > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench rc=$?"; cut -c1-400 $O/bench_$c.json
done
echo "== bench C3, Zipf(1.1) token pairs"; timeout 900 python bench.py --config C3 --zipf 1.1 --no-cpu --no-batch > $O/bench_C3zipf.json 2> $O/bench_C3zipf.err; echo "rc=$?"; cut -c1-300 $O/bench_C3zipf.json
cp $O/bench_C3.json $O/bench.json
if [ "$1" != quick ]; then
echo "== bench (pool-sharded path, one rank)"; timeout 900 python bench.py --force-dist --no-cpu > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?"
echo "== bench (two ranks sharing the GPU, one-shot exchange)"; timeout 900 python bench.py --gpus 2 --share-gpu --no-cpu > $O/bench_share2.json 2> $O/bench_share2.err; echo "rc=$?"; cut -c1-300 $O/bench_share2.json
echo "== upload"; timeout 300 python tools/upload_timing.py > $O/upload.json 2> $O/upload.err; cat $O/upload.json
echo "== small networks"; timeout 300 python tools/small_timing.py > $O/small.json 2> $O/small.err; cut -c1-600 $O/small.json
echo "== batched solves"; for c in C3 C4shard; do timeout 600 python tools/batch_timing.py --config $c > $O/batch_$c.jsonl 2> $O/batch_$c.err; cut -c1-300 $O/batch_$c.jsonl; done
echo "== the K-asset table's launch: 1e5 four-asset stableswap pools, warm and cold; 1e5 constant-sum"; timeout 300 python tools/profile_table.py > $O/table.jsonl 2> $O/table.err; CFMM_TABLE_WARM=0 timeout 300 python tools/profile_table.py >> $O/table.jsonl 2>> $O/table.err; timeout 300 python tools/profile_table.py --kind sum >> $O/table.jsonl 2>> $O/table.err; cut -c1-300 $O/table.jsonl
echo "== utility table (CFMM_ULOG) on 5e4 pools"; timeout 600 python tools/profile_ulog.py > $O/ulog.json 2> $O/ulog.err; cut -c1-600 $O/ulog.json
echo "== host share of a solve"; timeout 300 python tools/host_overhead.py 2> $O/host_overhead.err | head -1 > $O/host_overhead.json; cut -c1-400 $O/host_overhead.json
fi
cd /tmp
echo "== kernel trace of bench.py"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $O/trace_bench.log 2>&1; echo "rc=$?"
echo "== kernel trace of the K-asset table's launch and of the utility-table solve"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_table -o t -- python $R/tools/profile_table.py > $O/trace_table.log 2>&1; echo "rc=$?"; tail -1 $O/trace_table.log | cut -c1-300
CFMM_TABLE_WARM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_tablecold -o t -- python $R/tools/profile_table.py > $O/trace_tablecold.log 2>&1; echo "rc=$?"; tail -1 $O/trace_tablecold.log | cut -c1-300
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_table_SQ -o c -- python $R/tools/profile_table.py --launches 20 > $O/pmc_table_SQ.log 2>&1; echo "pmc table rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ulog -o t -- python $R/tools/profile_ulog.py > $O/trace_ulog.log 2>&1; echo "rc=$?"; tail -1 $O/trace_ulog.log | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_tablenewton -o t -- python $R/tools/profile_table_newton.py > $O/table_newton.jsonl 2> $O/trace_tablenewton.log; echo "rc=$?"; cut -c1-300 $O/table_newton.jsonl
echo "== kernel trace + vector-issue counters of the config-5 solve (second-order path)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_C5newton -o t -- python $R/tools/profile_newton.py --solves 3 > $O/trace_C5newton.log 2>&1; echo "rc=$?"; tail -1 $O/trace_C5newton.log | cut -c1-300
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_C5newton_SQ -o c -- python $R/tools/profile_newton.py --solves 1 > $O/pmc_C5newton_SQ.log 2>&1; echo "pmc C5newton rc=$?"
# the factorisation's side products run on the fp64 matrix pipe (chol2.hpp): instruction count and pipe-busy cycles
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/pmc_C5newton_MFMA -o c -- python $R/tools/profile_newton.py --solves 1 > $O/pmc_C5newton_MFMA.log 2>&1; echo "pmc C5newton MFMA rc=$?"
# PMC passes (each its own run).  SQ: instruction mix / issue cycles + the shader clock's cycles over the same dispatch (the
# effective clock under the profiler: bench.py's valu_frac divides by it, not by 2.4 GHz); WAIT: where the waves wait; TCC: L2 hit rate
PMC_FULL=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum")
PMC_BYTES=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_BUSY_CYCLES")
for cfg in C3 C4 C4x4 C3zipf C2 C5; do
  for tgt in eval iter; do
    [ $cfg = C5 ] && [ $tgt = iter ] && continue            # (config 5's outer iteration is the second-order one: traced above)
    base=$cfg; extra=""; [ $cfg = C3zipf ] && base=C3 && extra="--zipf 1.1"
    [ $tgt = eval ] && CMD="python $R/tools/profile_eval.py --config $base $extra" || CMD="python $R/tools/profile_iter.py --config $base $extra --solves 6"
    name=$cfg; [ $tgt = iter ] && name=${cfg}iter
    echo "== $name kernel trace"; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o t -- $CMD > $O/trace_$name.log 2>&1; echo "rc=$?"; tail -1 $O/trace_$name.log | cut -c1-300
    if [ $cfg = C3 ] || [ $cfg = C4 ]; then PMCS=("${PMC_FULL[@]}"); else PMCS=("${PMC_BYTES[@]}"); fi
    [ $cfg = C2 ] && PMCS=("FETCH_SIZE" "WRITE_SIZE")
    for pmc in "${PMCS[@]}"; do
      tag=$(echo $pmc | cut -d' ' -f1)
      timeout 900 rocprofv3 --pmc $pmc --output-format csv -d $O/pmc_${name}_$tag -o c -- $CMD > $O/pmc_${name}_$tag.log 2>&1; echo "pmc $name $tag rc=$?"
    done
  done
done
cd $R
# per-dispatch durations of the iteration / evaluation kernels (median / mean over the full launches) next to the stats tables
python - <<'PY'
import csv, glob, json
out = {}
for f in sorted(glob.glob('gpurun_out/p/trace_*/**/*kernel_trace.csv', recursive=True)):
    name = [p for p in f.split('/') if p.startswith('trace_')][0][6:]
    d = {}
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if any(s in k for s in ('iter_kernel', 'eval_kernel', 'update', 'eval_batch', 'solve_tiny', 'chol_', 'smooth_kernel', 'table_eval', 'gk_newton')):
            d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out[name] = {}
    for k, v in d.items():
        v.sort()
        ref = v[int(0.9 * (len(v) - 1))]                    # (the 90th percentile: ONE slow launch must not redefine what a full launch is)
        full = [x for x in v if x > 0.6 * ref]
        out[name][k] = dict(calls=len(v), median_us=v[len(v) // 2], mean_us=sum(v) / len(v), full_launches=len(full), full_mean_us=sum(full) / len(full), min_us=v[0], max_us=v[-1])
json.dump(out, open('gpurun_out/p/kernel_durations.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:2500])
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
du -sh $O
