#!/bin/bash
# Developer tool (GPU box): bench line + rocprofv3 per-kernel averages of a short bench run.
#   gpurun -- 'bash tools/quick_stats.sh gpurun_out/x'
cd "$(dirname "$0")/.."
ROOT=$(pwd); OUT=$ROOT/${1:-gpurun_out/quick}; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --no-cpu-baseline --no-secondary > $OUT/bench.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python $ROOT/bench.py --no-cpu-baseline --no-secondary --steps 20 > /dev/null 2>&1
python $ROOT/tools/rocprof_summary.py $OUT/stats | head -12 | cut -c1-140 > $OUT/kernel_stats.txt
rm -rf $OUT/stats
