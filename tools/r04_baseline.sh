#!/bin/bash
# Round-4 baseline on one box: the driver's bench command, the A/B of the in-kernel pair bias against the per-call cache
# (16-row kernels, same box), and the 32-row cached kernel.
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-r04_base} && mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.log 2> $OUT/bench_driver_cmd.err
tail -1 $OUT/bench_driver_cmd.log | cut -c1-900
for rep in 1 2 3; do
  python tools/bench_ipa.py 32 256 30 2>&1 | tail -1 | sed 's/^/in-kernel bias, 16-row one-block: /' >> $OUT/bias_ab.txt
  ABOPT_CORE32=0 python tools/bench_ipa_cached.py 32 256 5 2>&1 | tail -1 | sed 's/^/cached, 16-row persistent: /' >> $OUT/bias_ab.txt
  python tools/bench_ipa_cached.py 32 256 5 2>&1 | tail -1 | sed 's/^/cached, 32-row: /' >> $OUT/bias_ab.txt
done
cat $OUT/bias_ab.txt
