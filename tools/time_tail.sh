#!/bin/bash
# Developer tool: section clocks of one out_ln_mlp workgroup (OT_TIMING build) under optional ablation macros ($1: extra -D flags).
cd "$(dirname "$0")/.."
for a in "${@:-}"; do
  rm -f ab_opt_amd/csrc/mlp.o
  make -s -C ab_opt_amd/csrc CXXEXTRA="-DOT_TIMING $a" > /dev/null 2>&1 || { echo "build failed for $a"; continue; }
  echo "== flags: $a"
  python tools/bench_ipa_cached.py 32 256 4 2>&1 | grep -E "ot timing" | head -3
done
rm -f ab_opt_amd/csrc/mlp.o
make -s -C ab_opt_amd/csrc > /dev/null 2>&1
