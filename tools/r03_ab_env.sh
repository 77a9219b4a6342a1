#!/bin/bash
# Same-box A/B of one environment switch of the library (e.g. ABOPT_CORE32=1): rates interleaved, then kernel statistics under rocprofv3.
#   bash tools/r03_ab_env.sh <tag> <VAR=value> [run_shape args]
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-abenv} && mkdir -p $OUT
SW=$2; shift 2
ARGS=${@:---n 32 --l 256 --flavour abdesign --steps 20 --repeats 3}
export TMPDIR=/tmp
cd /tmp
for rep in 1 2; do
  python $ROOT/tools/run_shape.py $ARGS 2>/dev/null | sed "s/^/base: /" >> $OUT/rates.txt
  env $SW python $ROOT/tools/run_shape.py $ARGS 2>/dev/null | sed "s/^/$SW: /" >> $OUT/rates.txt
done
for name in base sw; do
  if [ $name = sw ]; then E=$SW; else E=ABOPT_NOOP=1; fi
  env $E rocprofv3 --kernel-trace --stats -d $OUT/st_$name -- python $ROOT/tools/run_shape.py $ARGS --repeats 1 > /dev/null 2>&1
  python $ROOT/tools/rocprof_summary.py $OUT/st_$name | head -12 | cut -c1-60,92-140 > $OUT/kernel_stats_$name.txt
  rm -rf $OUT/st_$name
done
cat $OUT/rates.txt; for f in $OUT/kernel_stats_*.txt; do echo "== $f"; cat $f; done
