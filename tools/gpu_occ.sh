#!/bin/bash
cd "$(dirname "$0")/.."
V=$PWD/cfmm-routing-code_amd/cfmm/variants
for cfg in C3 C4shard; do
echo -n "default  "; python tools/profile_eval.py --config $cfg | cut -c1-110
echo -n "nocurve  "; CFMM_LIB=$V/libcfmm_hip_nocurve.so python tools/profile_eval.py --config $cfg | cut -c1-110
done
