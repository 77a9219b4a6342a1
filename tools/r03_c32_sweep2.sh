#!/bin/bash
# Checks of the 32-row kernel's selection rule at its edges: the library's own choice against ABOPT_CORE32=0.
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-c32sweep2} && mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for shape in ${SHAPES:-23x256 28x256 62x256 32x200 36x224 40x256 18x256}; do
  set -- ${shape/x/ }
  for sw in auto 0; do
    if [ $sw = auto ]; then unset ABOPT_CORE32; else export ABOPT_CORE32=0; fi
    rocprofv3 --kernel-trace --stats -d $OUT/st -- python $ROOT/tools/run_shape.py --n $1 --l $2 --flavour abdesign --steps 6 --repeats 1 > /dev/null 2>&1
    echo "N=$1 L=$2 core32=$sw: $(python $ROOT/tools/rocprof_summary.py $OUT/st | grep 'ipa_core\|ipa_split' | cut -c1-40,92-130 | tr '\n' '|')" >> $OUT/sweep.txt
    rm -rf $OUT/st
  done
done
cat $OUT/sweep.txt
