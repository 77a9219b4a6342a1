#!/bin/bash
# Ablation timings of the 32-row core (variants built with -DC32_ABL=<bits>): average kernel time per variant under rocprofv3.
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-c32abl} && mkdir -p $OUT
shift
LIBS=$(for l in "$@"; do realpath $l; done)
export TMPDIR=/tmp ABOPT_CORE32=1
cd /tmp
for lib in $ROOT/ab_opt_amd/libabopt_hip.so $LIBS; do
  name=$(basename $lib .so)
  ABOPT_LIB_PATH=$lib rocprofv3 --kernel-trace --stats -d $OUT/st_$name -- python $ROOT/tools/run_shape.py --n 32 --l 256 --flavour abdesign --steps 10 --repeats 1 > /dev/null 2>&1
  echo "$name: $(python $ROOT/tools/rocprof_summary.py $OUT/st_$name | grep core32 | cut -c1-40,60-130)" >> $OUT/abl.txt
  rm -rf $OUT/st_$name
done
cat $OUT/abl.txt
