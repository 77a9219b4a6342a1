#!/bin/bash
# Developer tool: A/B of the persistent IPA core against the one-block-per-workgroup kernel on the same box.
cd "$(dirname "$0")/.."
for v in "" "-DCORE_NO_PERSIST" "" "-DCORE_NO_PERSIST"; do
  rm -f ab_opt_amd/csrc/ipa_core.o
  make -s -C ab_opt_amd/csrc CXXEXTRA="$v" > /dev/null 2>&1 || { echo "build failed"; continue; }
  echo -n "[$v] N=32: "; python tools/bench_ipa_cached.py 32 256 6 2>&1 | tail -1
  echo -n "[$v] N=64: "; python tools/bench_ipa_cached.py 64 256 6 2>&1 | tail -1
done
rm -f ab_opt_amd/csrc/ipa_core.o; make -s -C ab_opt_amd/csrc > /dev/null 2>&1
