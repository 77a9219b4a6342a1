#!/bin/bash
cd "$(dirname "$0")/../.." && ROOT=$(pwd)
Z=ABOPT_DEV_ZTERMS=1
ROUNDS=3 bash tools/r06/ab_graph.sh hx_abg2 base hx1,$Z hx7,$Z 2>&1 | tail -10
echo "== two streams, base"; python tools/r03_two_streams.py --n 32 --steps 20 2>&1 | tail -1
echo "== two streams, base, offset"; python tools/r03_two_streams.py --n 32 --steps 20 --offset 150000 2>&1 | tail -1
echo "== two streams, hx7"; ABOPT_LIB_PATH=$ROOT/ab_opt_amd/variants/libabopt_hx7.so ABOPT_DEV_ZTERMS=1 python tools/r03_two_streams.py --n 32 --steps 20 2>&1 | tail -1
echo "== two streams, hx7, offset"; ABOPT_LIB_PATH=$ROOT/ab_opt_amd/variants/libabopt_hx7.so ABOPT_DEV_ZTERMS=1 python tools/r03_two_streams.py --n 32 --steps 20 --offset 150000 2>&1 | tail -1
