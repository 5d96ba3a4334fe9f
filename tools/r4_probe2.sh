cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_newton.py -m gpu -q -x -k "token_limit or refuses" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
(time timeout 900 python bench.py --config C4x4 --no-cpu --steps 3 --warmup 1) > $O/bench_C4x4.json 2> $O/bench_C4x4.err; echo "rc=$?"; cut -c1-1500 $O/bench_C4x4.json; tail -4 $O/bench_C4x4.err
for pp in 0 1; do CFMM_PINGPONG=$pp timeout 600 python tools/microbench.py --config C4x4 --tag pp$pp --solves 3 --reps 20 2>> $O/mb.err | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['tag'], r['status'], r['evals'], 'dev_us/eval %.2f eval_all_us %.2f' % (r['dev_us_per_eval'], r['eval_all_us']))"; done
timeout 900 python bench.py --config C3 --zipf 1.1 --no-cpu --no-batch --steps 5 > $O/bench_C3zipf.json 2> $O/bench_C3zipf.err; echo "rc=$?"; cut -c1-700 $O/bench_C3zipf.json; tail -3 $O/bench_C3zipf.err
