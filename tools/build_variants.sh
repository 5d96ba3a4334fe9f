#!/bin/bash
# builds the kernel-tuning variants of libcfmm_hip.so (cross-compiles; no GPU needed)
set -e
cd "$(dirname "$0")/../cfmm-routing-code_amd/csrc"
rm -f ../cfmm/variants/*.so
build() { make -s variant TAG=$1 DEFS="$2" & }
build t1024w4 "-DEVAL_THREADS_DEF=1024 -DEVAL_WAVES_PER_SIMD=4"
build t640w5 "-DEVAL_THREADS_DEF=640 -DEVAL_WAVES_PER_SIMD=5"
build t768w6 "-DEVAL_THREADS_DEF=768 -DEVAL_WAVES_PER_SIMD=6"
build t512w6 "-DEVAL_THREADS_DEF=512 -DEVAL_WAVES_PER_SIMD=6"
build t384w6 "-DEVAL_THREADS_DEF=384 -DEVAL_WAVES_PER_SIMD=6"
build timers "-DCFMM_PHASE_TIMERS"
wait
ls ../cfmm/variants/
