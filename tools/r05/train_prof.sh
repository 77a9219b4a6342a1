#!/bin/bash
# Steady-state kernel statistics of a training step (config 5): two rocprofv3 runs (3 and 7 steps), differenced.  bash tools/r05/train_prof.sh <tag>
cd "$(dirname "$0")/../.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-r05_train} && mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for n in 3 7; do STEPS=$n rocprofv3 --kernel-trace --stats -d $OUT/st$n -- python $ROOT/tools/prof_train_full.py > $OUT/log$n.txt 2>&1; done
cd $ROOT
python tools/r05/train_steady.py $OUT/st3 $OUT/st7 3 7 > $OUT/train_steady_step_kernel_stats.txt
rm -rf $OUT/st3 $OUT/st7
head -45 $OUT/train_steady_step_kernel_stats.txt | cut -c1-80,92-140
