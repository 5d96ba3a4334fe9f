#!/bin/bash
# builds the kernel-tuning variants of libcfmm_hip.so (cross-compiles; no GPU needed)
set -e
cd "$(dirname "$0")/../cfmm-routing-code_amd/csrc"
rm -f ../cfmm/variants/*.so
build() { make -s variant TAG=$1 DEFS="$2" & }
build t1024w4 "-DEVAL_THREADS_DEF=1024 -DEVAL_WAVES_PER_SIMD=4"
build t512w4 "-DEVAL_THREADS_DEF=512 -DEVAL_WAVES_PER_SIMD=4"
build t512w2 "-DEVAL_THREADS_DEF=512 -DEVAL_WAVES_PER_SIMD=2"
build t256w4 "-DEVAL_THREADS_DEF=256 -DEVAL_WAVES_PER_SIMD=4"
build t1024w6 "-DEVAL_THREADS_DEF=1024 -DEVAL_WAVES_PER_SIMD=6"
build timers "-DCFMM_PHASE_TIMERS"
wait
ls ../cfmm/variants/
