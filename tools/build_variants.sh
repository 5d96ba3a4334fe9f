#!/bin/bash
# builds the kernel-tuning variants of libcfmm_hip.so (cross-compiles; no GPU needed)
set -e
cd "$(dirname "$0")/../cfmm-routing-code_amd/csrc"
build() { make -s variant TAG=$1 DEFS="$2" & }
build t512w2 "-DEVAL_THREADS_DEF=512 -DEVAL_WAVES_PER_SIMD=2"
build t512w4 "-DEVAL_THREADS_DEF=512 -DEVAL_WAVES_PER_SIMD=4"
build t256w2 "-DEVAL_THREADS_DEF=256 -DEVAL_WAVES_PER_SIMD=2"
build t256w3 "-DEVAL_THREADS_DEF=256 -DEVAL_WAVES_PER_SIMD=3"
build t256w4 "-DEVAL_THREADS_DEF=256 -DEVAL_WAVES_PER_SIMD=4"
wait
ls -la ../cfmm/variants/
