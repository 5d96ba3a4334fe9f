cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_newton.py -m gpu -q -x 2>&1 | tail -8
for b in inverse classic; do CFMM_BACKSUB=$b timeout 600 python tools/profile_newton.py --solves 5 2>&1 | tail -1 | cut -c1-600; CFMM_BACKSUB=$b python tools/chol_probe.py 2>&1 | tail -2; done
