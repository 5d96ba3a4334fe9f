#!/bin/bash
cd "$(dirname "$0")/.."
V=$PWD/cfmm-routing-code_amd/cfmm/variants
for lib in 409dfa9 922b710 default 409dfa9 922b710 default; do
  L=$V/libcfmm_hip_$lib.so; [ $lib = default ] && L=
  echo -n "$lib: "; CFMM_LIB=$L python tools/kernel_budget.py --only C3 C2 --rounds 4 | tail -1 | cut -c1-330
done
