#!/bin/bash
# Same-box timings of library variants (developer builds under ab_opt_amd/variants/) with rocprofv3: average time of the 32-row kernel(s) per variant,
# two rounds, interleaved.   bash tools/r05/abl.sh <tag> <lib or tag of ab_opt_amd/variants/libabopt_<tag>.so>[,ENV=VAL[,ENV=VAL]] ...     ("base" = the product library)
cd "$(dirname "$0")/../.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-abl} && mkdir -p $OUT
shift
export TMPDIR=/tmp ABOPT_CORE32=1
cd /tmp
: > $OUT/abl.txt
for round in 1 2; do
for spec in "$@"; do
  IFS=',' read -r -a parts <<< "$spec"
  lib=${parts[0]}
  if [ "$lib" = base ]; then lib=$ROOT/ab_opt_amd/libabopt_hip.so; elif [ ! -f "$lib" ]; then lib=$ROOT/ab_opt_amd/variants/libabopt_$lib.so; fi
  envs=("${parts[@]:1}")
  name=$(echo "$spec" | tr ',/' '__')
  env ABOPT_LIB_PATH=$lib "${envs[@]}" rocprofv3 --kernel-trace --stats -d $OUT/st_$name -- python $ROOT/tools/run_shape.py --n ${ABL_N:-32} --l ${ABL_L:-256} --flavour abdesign --steps 10 --repeats 1 > $OUT/log_$name.txt 2>&1
  echo "$spec: $(python $ROOT/tools/rocprof_summary.py $OUT/st_$name | grep -E "core32|node_frags|out_ln_mlp" | cut -c1-46,60-118 | tr '\n' '|')" >> $OUT/abl.txt
  rm -rf $OUT/st_$name
done; done
cat $OUT/abl.txt
