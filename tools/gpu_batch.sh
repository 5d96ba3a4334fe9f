#!/bin/bash
# batched solves (cfmm_solve_batch): parity tests, then the cost of a lock-step iteration against B
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/batch; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "${TESTS:-batched or clones}" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-400
for CFG in ${CONFIGS:-C3 C4shard}; do
  timeout 600 python tools/batch_timing.py --config $CFG > $O/batch_$CFG.jsonl 2> $O/batch_$CFG.err; echo "rc=$?"; cat $O/batch_$CFG.jsonl | cut -c1-600; tail -3 $O/batch_$CFG.err
done
if [ -n "$PROFILE" ]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_batch -o t -- python $R/tools/batch_timing.py --config C3 --sizes 8 --reps 3 > $O/trace_batch.log 2>&1; echo "rc=$?"
  cd $R
  f=$(find $O/trace_batch -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-200
  find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
fi
