#!/bin/bash
# config 5 / the dense factorisation: blocked diagonal factor (tree) against the one-wave 32-column elimination (variant), tests first
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
V=$PWD/cfmm-routing-code_amd/cfmm/variants

for rep in 1 2; do
  for lib in default unblocked; do
    L=$V/libcfmm_hip_$lib.so; [ "$lib" = default ] && L=
    echo "== $lib"; CFMM_LIB=$L python tools/chol_probe.py 2>&1 | tail -3; CFMM_LIB=$L python tools/c5_quick.py 2>&1 | tail -2
  done
done
