#!/bin/bash
# Same-box A/B of two builds of the library (ABOPT_LIB_PATH): rates interleaved, then kernel statistics under rocprofv3.
#   bash tools/r03_ab_lib.sh <tag> <libA> <libB> [run_shape args]
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-ab} && mkdir -p $OUT
A=$2; B=$3; shift 3
ARGS=${@:---n 32 --l 256 --flavour abdesign --steps 20 --repeats 3}
export TMPDIR=/tmp
cd /tmp
for rep in 1 2; do for lib in $A $B; do
  name=$(basename $lib .so)
  ABOPT_LIB_PATH=$lib python $ROOT/tools/run_shape.py $ARGS 2>/dev/null | sed "s/^/$name: /" >> $OUT/rates.txt
done; done
for lib in $A $B; do
  name=$(basename $lib .so)
  ABOPT_LIB_PATH=$lib rocprofv3 --kernel-trace --stats -d $OUT/st_$name -- python $ROOT/tools/run_shape.py $ARGS --repeats 1 > /dev/null 2>&1
  python $ROOT/tools/rocprof_summary.py $OUT/st_$name | head -12 | cut -c1-60,92-140 > $OUT/kernel_stats_$name.txt
  rm -rf $OUT/st_$name
done
cat $OUT/rates.txt; for f in $OUT/kernel_stats_*.txt; do echo "== $f"; cat $f; done
