#!/bin/bash
# same-box A/B of one environment switch (0 / 1) on the replayed loop, alternating, several shapes:   bash tools/r06/ab_env.sh ABOPT_X_TERMS
cd "$(dirname "$0")/../.." && ROOT=$(pwd); VAR=$1
for round in 1 2 3; do
for args in "--n 32 --l 256 --flavour abdesign" "--n 1000 --l 48 --shared --flavour abdock" "--n 8 --l 256 --flavour abdesign"; do
  for f in 0 1; do
    echo "$VAR=$f $args: $(env $VAR=$f python tools/run_shape.py $args --steps 20 --repeats 5 --graph 2>&1 | grep shape | sed 's/.*: //')"
  done
done; done
