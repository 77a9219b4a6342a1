#!/bin/bash
# Same-box A/B of one environment knob: step rates interleaved, then kernel statistics under rocprofv3.
#   bash tools/r04_ab_env.sh <tag> <ENVVAR> [run_shape args]      e.g.  r04_ab_env.sh xt ABOPT_X_TERMS
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-ab_env} && mkdir -p $OUT
VAR=$2
shift; shift
ARGS=${@:---n 32 --l 256 --flavour abdesign --steps 20 --repeats 5 --graph}
export TMPDIR=/tmp
cd /tmp
: > $OUT/rates.txt
for rep in 1 2 3; do for f in 0 1; do
  env $VAR=$f python $ROOT/tools/run_shape.py $ARGS 2>/dev/null | sed "s/^/$VAR=$f: /" >> $OUT/rates.txt
done; done
for f in 0 1; do
  env $VAR=$f rocprofv3 --kernel-trace --stats -d $OUT/st_$f -- python $ROOT/tools/run_shape.py ${ARGS/--graph/} --repeats 1 > /dev/null 2>&1
  python $ROOT/tools/rocprof_summary.py $OUT/st_$f | head -12 | cut -c1-60,92-140 > $OUT/kernel_stats_$f.txt
  rm -rf $OUT/st_$f
done
cat $OUT/rates.txt; for f in $OUT/kernel_stats_*.txt; do echo "== $f"; cat $f; done
