#!/bin/bash
# round 2, first GPU check: parity suite, bench lines (C3, pool-sharded path with one rank, C4 strong config), upload timing
cd "$(dirname "$0")/.."
O=gpurun_out/r2a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== gpu tests"; timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
echo "== bench C3"; timeout 600 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; echo "rc=$?"; cut -c1-1500 $O/bench_c3.json; tail -3 $O/bench_c3.err
echo "== bench C3 --force-dist"; timeout 600 python bench.py --force-dist --no-cpu > $O/bench_c3_dist.json 2> $O/bench_c3_dist.err; echo "rc=$?"; cut -c1-1200 $O/bench_c3_dist.json; tail -3 $O/bench_c3_dist.err
echo "== bench C4"; timeout 600 python bench.py --config C4 --no-cpu --steps 5 > $O/bench_c4.json 2> $O/bench_c4.err; echo "rc=$?"; cut -c1-1200 $O/bench_c4.json; tail -3 $O/bench_c4.err
echo "== upload timing"; timeout 300 python tools/upload_timing.py > $O/upload.json 2> $O/upload.err; cat $O/upload.json; tail -3 $O/upload.err
