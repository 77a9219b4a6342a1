#!/bin/bash
# Training step (config 5): kernel statistics of four full steps under rocprofv3 + share of GPU time in abopt:: kernels.
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-train} && mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/st -- python $ROOT/tools/prof_train_full.py > $OUT/log.txt 2>&1
cd $ROOT
python tools/rocprof_summary.py $OUT/st --all > $OUT/kernel_stats.txt
rm -rf $OUT/st
head -30 $OUT/kernel_stats.txt | cut -c1-70,92-140
