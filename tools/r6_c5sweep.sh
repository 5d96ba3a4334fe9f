#!/bin/bash
# config 5: solve time against the second-order knobs (prelude length, initial barrier weight, shrink factor, chord steps)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for pre in 12 8 6 4 0; do CFMM_NEWTON_PRELUDE=$pre python tools/c5_quick.py 2>&1 | tail -2; done
for mu0 in 0.03 0.3 1.0; do CFMM_NEWTON_MU0=$mu0 python tools/c5_quick.py 2>&1 | tail -2; done
for ch in 0 5; do echo "chord $ch"; CFMM_CHORD=$ch python tools/c5_quick.py 2>&1 | tail -2; done
