#!/bin/bash
# C3 / C2 kernel times of saved builds of two commits against the working tree, interleaved twice (box noise shows as the
# spread between the two rounds).  The saved builds: for each commit  git worktree add /tmp/wt/<sha> <sha>;  make -C
# /tmp/wt/<sha>/cfmm-routing-code_amd/csrc;  copy its libcfmm_hip.so to cfmm/variants/libcfmm_hip_<sha>.so  (not kept in the tree).
cd "$(dirname "$0")/.."
V=$PWD/cfmm-routing-code_amd/cfmm/variants
for lib in 409dfa9 922b710 default 409dfa9 922b710 default; do
  L=$V/libcfmm_hip_$lib.so; [ $lib = default ] && L=
  echo -n "$lib: "; CFMM_LIB=$L python tools/kernel_budget.py --only C3 C2 --rounds 4 | tail -1 | cut -c1-330
done
