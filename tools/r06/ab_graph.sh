#!/bin/bash
# same-box A/B of library variants on the replayed K-step loop (what bench.py times): ms per step, alternating, ROUNDS rounds
#   bash tools/r06/ab_graph.sh <out tag> <lib tag | base>[,ENV=VAL...] ...
cd "$(dirname "$0")/../.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-abg} && mkdir -p $OUT; shift
: > $OUT/ab.txt
for round in $(seq 1 ${ROUNDS:-3}); do
for spec in "$@"; do
  IFS=',' read -r -a parts <<< "$spec"
  lib=${parts[0]}
  if [ "$lib" = base ]; then lib=$ROOT/ab_opt_amd/libabopt_hip.so; else lib=$ROOT/ab_opt_amd/variants/libabopt_$lib.so; fi
  envs=("${parts[@]:1}")
  echo "$spec: $(env ABOPT_LIB_PATH=$lib "${envs[@]}" python $ROOT/tools/run_shape.py --n ${ABL_N:-32} --l ${ABL_L:-256} --flavour abdesign --steps 20 --repeats 5 --graph 2>&1 | grep shape)" >> $OUT/ab.txt
done; done
cat $OUT/ab.txt
