#!/bin/bash
# Second pass of a round's artefacts: the bench lines again, now citing the PMC / trace summaries that tools/collect_profiles.py has
# just put under profiles/ (bench.py reads the newest ones at run time), + the kernel-time budget.  Results -> gpurun_out/p/ (merged over
# the first pass's files), gpurun_out/b/budget.json.
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/p; mkdir -p $O $R/gpurun_out/b
export TMPDIR=/tmp
for c in C3 C2 C4 C5 C4x4; do
  timeout 1200 python bench.py --config $c --cpu-seconds 8 > $O/bench_$c.json 2> $O/bench_$c.err; echo "bench $c rc=$?"; cut -c1-200 $O/bench_$c.json
done
timeout 900 python bench.py --config C3 --zipf 1.1 --no-cpu --no-batch > $O/bench_C3zipf.json 2> $O/bench_C3zipf.err; echo "zipf rc=$?"
cp $O/bench_C3.json $O/bench.json
timeout 900 python bench.py --force-dist --no-cpu > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "dist1 rc=$?"
timeout 900 python bench.py --gpus 2 --share-gpu --no-cpu > $O/bench_share2.json 2> $O/bench_share2.err; echo "share2 rc=$?"
if [ "$1" = budget ]; then python tools/kernel_budget.py --write > $R/gpurun_out/b/write.log 2>&1; cp profiles/budget.json $R/gpurun_out/b/budget.json; tail -1 $R/gpurun_out/b/write.log | cut -c1-600; fi
