#!/bin/bash
# Round 3: kernel statistics of the shapes other than the bench's (reference headline N=1000 poses of one L~48 crop; small batches).
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-shapes} && mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for spec in "1000 48 --shared abdock" "8 256 _ abdesign" "2 256 _ abdesign" "64 256 --shared abdock"; do
  set -- $spec; n=$1; l=$2; sh=$3; fl=$4; [ "$sh" = "_" ] && sh=""
  tag=n${n}_l${l}
  python $ROOT/tools/run_shape.py --n $n --l $l $sh --flavour $fl --steps 10 >> $OUT/rates.txt 2>> $OUT/err.txt
  python $ROOT/tools/run_shape.py --n $n --l $l $sh --flavour $fl --steps 10 --graph >> $OUT/rates.txt 2>> $OUT/err.txt
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$tag -- python $ROOT/tools/run_shape.py --n $n --l $l $sh --flavour $fl --steps 10 --repeats 1 > $OUT/prof_$tag.log 2>&1
  python $ROOT/tools/rocprof_summary.py $OUT/stats_$tag | head -16 > $OUT/kernel_stats_$tag.txt
  rm -rf $OUT/stats_$tag
done
cat $OUT/rates.txt
