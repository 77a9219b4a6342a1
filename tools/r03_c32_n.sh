#!/bin/bash
# Core time of the 32-row kernel against the batch size (fewer workgroups than CUs: is a block's speed set by the CU or by the chip's HBM?)
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-c32n} && mkdir -p $OUT
export TMPDIR=/tmp ABOPT_CORE32=1 ABOPT_CORE_NO_SPLIT=1
cd /tmp
for n in 8 16 24 32 64; do
  rocprofv3 --kernel-trace --stats -d $OUT/st_$n -- python $ROOT/tools/run_shape.py --n $n --l 256 --flavour abdesign --steps 10 --repeats 1 > /dev/null 2>&1
  echo "N=$n: $(python $ROOT/tools/rocprof_summary.py $OUT/st_$n | grep 'core32\|out_ln\|node_frags' | cut -c1-40,92-130 | tr '\n' '|')" >> $OUT/n.txt
  rm -rf $OUT/st_$n
done
cat $OUT/n.txt
