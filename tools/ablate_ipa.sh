#!/bin/bash
# Developer tool: rebuild the IPA core with each ablation mask (see CORE_ABL in csrc/ipa_core.hip) and time it at the bench shape.
#   gpurun -- 'bash tools/ablate_ipa.sh "0 1 2 4" > gpurun_out/ablate.log'
cd "$(dirname "$0")/.."
for a in ${1:-0 1 2 3 4 8 16 28 32 63}; do
  rm -f ab_opt_amd/csrc/ipa_core.o
  make -s -C ab_opt_amd/csrc CXXEXTRA=-DCORE_ABL=$a > /dev/null 2>&1 || { echo "build failed for $a"; continue; }
  echo -n "ABL=$a: "
  python tools/bench_ipa_cached.py ${2:-32} ${3:-256} 4 2>&1 | tail -1
done
rm -f ab_opt_amd/csrc/ipa_core.o
make -s -C ab_opt_amd/csrc > /dev/null 2>&1
