#!/bin/bash
# pair_embed_kernel with 64-wide products on the 16-bit matrix pipe (developer builds -DPE_TERMS=<mask>; three bf16 terms until round 5, two fp16 terms since): correctness (encode tests + the 200-repeat race
# detector) and time of each variant on one box.   bash tools/r05/pe_terms.sh <tag> <mask> [<mask> ...]
cd "$(dirname "$0")/../.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-pe} && mkdir -p $OUT; shift
: > $OUT/summary.txt
for v in base "$@"; do
  lib=$ROOT/ab_opt_amd/variants/libabopt_pe$v.so; [ $v = base ] && lib=$ROOT/ab_opt_amd/libabopt_hip.so
  t=$(ABOPT_LIB_PATH=$lib python tools/bench_encode.py 32 256 10 2>&1 | tail -1)
  r=$(ABOPT_LIB_PATH=$lib python -m pytest tests -m gpu -q -x -k "encode or pair_embedding_repeats or pair_embed_backward" 2>&1 | tail -1)
  echo "PE_TERMS=$v | $t | $r" >> $OUT/summary.txt
done
cat $OUT/summary.txt
