cd "$(dirname "$0")/.."
O=gpurun_out/r4c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_newton.py -m gpu -q -x -k "token_limit or refuses" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
timeout 900 python tools/kernel_budget.py --write > $O/budget_write.json 2> $O/budget_write.err; echo "write rc=$?"; cat $O/budget_write.json; tail -3 $O/budget_write.err
timeout 900 python tools/kernel_budget.py > $O/budget_check.json 2> $O/budget_check.err; echo "check rc=$? (0 expected)"; cut -c1-300 $O/budget_check.json
CFMM_LIB=$PWD/cfmm-routing-code_amd/cfmm/variants/libcfmm_hip_misaligned.so timeout 900 python tools/kernel_budget.py --only C3 > $O/budget_misaligned.json 2> $O/budget_misaligned.err; echo "misaligned rc=$? (1 expected)"; cat $O/budget_misaligned.json
cp profiles/budget.json $O/budget.json
timeout 900 python -m pytest tests/test_gpu_perf.py -m gpu -q > $O/pytest_perf.log 2>&1; echo "perf pytest rc=$?"; tail -3 $O/pytest_perf.log
