cd "$(dirname "$0")/.."
O=gpurun_out/r4d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_table.py -m gpu -q -x > $O/pytest_table.log 2>&1; echo "table rc=$?"; tail -30 $O/pytest_table.log | cut -c1-400
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_table.py > $O/pytest.log 2>&1; echo "all rc=$?"; tail -5 $O/pytest.log | cut -c1-300
