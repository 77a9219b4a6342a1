#!/bin/bash
# Developer tool: section timers of one IPA-core workgroup (CORE_TIMING build), optionally combined with ablation masks.
cd "$(dirname "$0")/.."
for a in ${1:-0}; do
  rm -f ab_opt_amd/csrc/ipa_core.o
  make -s -C ab_opt_amd/csrc CXXEXTRA="-DCORE_TIMING -DCORE_ABL=$a $EXTRA" > /dev/null 2>&1 || { echo "build failed for $a"; continue; }
  echo "== ABL=$a"
  python tools/bench_ipa_cached.py ${2:-32} ${3:-256} 4 2>&1 | grep -E "core timing|cached path"
done
rm -f ab_opt_amd/csrc/ipa_core.o
make -s -C ab_opt_amd/csrc > /dev/null 2>&1
