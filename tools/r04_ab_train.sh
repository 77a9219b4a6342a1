#!/bin/bash
# Same-box A/B of one environment knob on the training step (tools/r04_train_ms.py) and the GEMM shape table: r04_ab_train.sh <tag> <ENVVAR>
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-ab_train} && mkdir -p $OUT
VAR=$2
cd /tmp; export TMPDIR=/tmp
: > $OUT/train.txt
for rep in 1 2 3; do for f in 0 1; do env $VAR=$f python $ROOT/tools/r04_train_ms.py 10 2>/dev/null | sed "s/^/$VAR=$f: /" >> $OUT/train.txt; done; done
for f in 0 1; do env $VAR=$f python $ROOT/tools/r04_gemm_shapes.py 2>/dev/null > $OUT/shapes_$f.txt; done
cat $OUT/train.txt; paste -d'|' <(cut -c1-52,58-68 $OUT/shapes_0.txt) <(cut -c58-68 $OUT/shapes_1.txt)
