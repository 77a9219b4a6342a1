#!/bin/bash
# Developer tool: same-box A/B of one compile-time flag on one kernel.   bash tools/ab_flag.sh <file.o stem> <-DFLAG> <kernel name substring>
cd "$(dirname "$0")/.."
ROOT=$(pwd)
for v in "" "$2" "" "$2"; do
  rm -f ab_opt_amd/csrc/$1.o
  make -s -C ab_opt_amd/csrc CXXEXTRA="$v" > /dev/null 2>&1 || { echo "build failed [$v]"; continue; }
  export TMPDIR=/tmp
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/abt -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 20 > /dev/null 2>&1)
  echo "[$v] $(python tools/rocprof_summary.py /tmp/abt | grep "$3" | head -1 | cut -c1-140)"
  rm -rf /tmp/abt
done
rm -f ab_opt_amd/csrc/$1.o; make -s -C ab_opt_amd/csrc > /dev/null 2>&1
