#!/bin/bash
# same-box A/B of library variants on the replayed loop at several shapes, alternating:  bash tools/r06/ab_lib.sh base cv1
cd "$(dirname "$0")/../.." && ROOT=$(pwd)
for round in 1 2 3; do
for args in "--n 32 --l 256 --flavour abdesign" "--n 64 --l 256 --shared --flavour abdock" "--n 1000 --l 48 --shared --flavour abdock"; do
  for v in "$@"; do
    lib=$ROOT/ab_opt_amd/variants/libabopt_$v.so; [ $v = base ] && lib=$ROOT/ab_opt_amd/libabopt_hip.so
    echo "$v $args: $(ABOPT_LIB_PATH=$lib python tools/run_shape.py $args --steps 20 --repeats 5 --graph 2>&1 | grep shape | sed 's/.*: //')"
  done
done; done
