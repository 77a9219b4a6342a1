#!/bin/bash
# second pass of the K-packed fp16 experiment: two-launch core (ABOPT_FUSE_TAIL=0) of base / hx1 / hx7 with and without the z + bias streams (-DC32_ABL=1), and the role stamps
cd "$(dirname "$0")/../.." && ROOT=$(pwd)
Z=ABOPT_DEV_ZTERMS=1; U=ABOPT_FUSE_TAIL=0
ABL_N=32 ABL_L=256 bash tools/r05/abl.sh hx_abl2 base,$U hx1,$U,$Z hx7,$U,$Z hx0a,$U hx1a,$U,$Z hx7a,$U,$Z 2>&1 | tail -14
for v in hx0t hx1t hx7t; do
  echo "== $v"; ABOPT_LIB_PATH=$ROOT/ab_opt_amd/variants/libabopt_$v.so ABOPT_FUSE_TAIL=0 ABOPT_DEV_ZTERMS=1 ABOPT_CORE32=1 python tools/run_shape.py --n 32 --l 256 --flavour abdesign --steps 4 --repeats 1 2>&1 | grep -E "core32 timing|shape"
done
