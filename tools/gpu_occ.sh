#!/bin/bash
cd "$(dirname "$0")/.."
V=$PWD/cfmm-routing-code_amd/cfmm/variants
python -m pytest tests -m gpu -q -x 2>&1 | tail -2
for cfg in C3 C4shard; do
for lib in t1024w4 t640w5 t768w6 t512w6 t384w6; do
echo -n "$lib "; CFMM_LIB=$V/libcfmm_hip_$lib.so python tools/profile_eval.py --config $cfg | cut -c1-112
done; done
