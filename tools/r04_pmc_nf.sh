#!/bin/bash
# HBM bytes of node_frags under the two workgroup -> (row group, head) deals (ABOPT_NF_MAP=0 head-major 2-D grid | 1 units of 4 heads per XCD)
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-pmc_nf} && mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/bench.py --steps 4 --warmup 1 --repeats 1 --graph off --no-prof --no-cpu-baseline --no-secondary"
for m in 0 1; do
  i=0
  for grp in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    ABOPT_NF_MAP=$m rocprofv3 --kernel-trace --pmc $grp -d $OUT/m$m/g$i --output-format csv -- $CMD > $OUT/m${m}_g$i.log 2>&1
  done
  python $ROOT/tools/pmc_digest.py $OUT/m$m --kernel node_frags_kernel > $OUT/nf_map$m.txt 2>&1
  rm -rf $OUT/m$m
  head -8 $OUT/nf_map$m.txt
done
