#!/bin/bash
# Where does the 32-row core pay?  Core kernel time per launch with and without ABOPT_CORE32=1 over batch sizes / lengths.
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-c32sweep} && mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for shape in "16 256" "24 256" "32 256" "48 256" "64 256" "64 128" "128 128" "256 64" "20 400"; do
  set -- $shape
  for sw in 0 1; do
    ABOPT_CORE32=$sw rocprofv3 --kernel-trace --stats -d $OUT/st -- python $ROOT/tools/run_shape.py --n $1 --l $2 --flavour abdesign --steps 6 --repeats 1 > /dev/null 2>&1
    echo "N=$1 L=$2 core32=$sw: $(python $ROOT/tools/rocprof_summary.py $OUT/st | grep 'ipa_core\|ipa_split' | cut -c1-40,92-130 | tr '\n' '|')" >> $OUT/sweep.txt
    rm -rf $OUT/st
  done
done
cat $OUT/sweep.txt
