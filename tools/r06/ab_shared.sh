#!/bin/bash
# pair terms on / off at shapes whose pair features are shared (one complex, N samples: the reference's own batches) -- ms per step, replayed loop, alternating
cd "$(dirname "$0")/../.." && ROOT=$(pwd)
for round in 1 2; do
for args in "--n 64 --l 256 --shared --flavour abdock" "--n 32 --l 256 --shared --flavour abdesign" "--n 128 --l 256 --shared --flavour abdesign" "--n 1000 --l 48 --shared --flavour abdock"; do
  for f in 0 1; do
    echo "terms=$f $args: $(ABOPT_PAIR_TERMS=$f python tools/run_shape.py $args --steps 20 --repeats 5 --graph 2>&1 | grep shape | sed 's/.*: //')"
  done
done; done
