cd "$(dirname "$0")/.."
for pre in 4 8 12 16; do CFMM_NEWTON_PRELUDE=$pre python tools/c5_quick.py 2>&1 | tail -2; done
for mu0 in 0.03 0.3 1.0; do CFMM_NEWTON_MU0=$mu0 python tools/c5_quick.py 2>&1 | tail -2; done
