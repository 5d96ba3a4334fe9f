#!/bin/bash
# explicit second-order solves with SIX processes sharing the device: the in-place diagonal block of the pair Cholesky must not be read late
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/stress
for round in 1 2 3; do
  pids=()
  for i in 1 2 3 4 5 6; do
    timeout 300 python tools/newton_seeds.py 1171 1262 1463 1171 1262 1463 > gpurun_out/stress/p${round}_$i.txt 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
done
echo "newton lines: $(cat gpurun_out/stress/p*.txt | grep -c '   newton:')  not optimal: $(cat gpurun_out/stress/p*.txt | grep '   newton:' | grep -vc optimal)  raised: $(cat gpurun_out/stress/p*.txt | grep -c raised)"
cat gpurun_out/stress/p*.txt | grep '   newton:' | grep -v optimal | head -5
timeout 900 python -m pytest tests/test_gpu_newton.py -m gpu -x -q 2>&1 | tail -3
