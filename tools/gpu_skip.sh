#!/bin/bash
cd "$(dirname "$0")/.."
V=$PWD/cfmm-routing-code_amd/cfmm/variants
for cfg in C3 C4shard C2; do
CFMM_LIB=$V/libcfmm_hip_skip.so python tools/profile_eval.py --config $cfg
python tools/profile_eval.py --config $cfg
done
