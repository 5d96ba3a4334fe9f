cd "$(dirname "$0")/.."
for c in 0 3; do echo "== CFMM_CHORD=$c"; CFMM_CHORD=$c CFMM_NEWTON_TRACE=1 python tools/profile_newton.py --solves 4 2>&1 | grep -E "steps, |solves" | tail -3 | cut -c1-1500; done
timeout 900 python -m pytest tests/test_gpu_newton.py tests/test_gpu.py tests/test_cvx.py -m gpu -q 2>&1 | tail -5
