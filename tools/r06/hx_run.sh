#!/bin/bash
# Timing experiment of VERDICT r05 item 1: the 32-row kernel's aggregations on K-packed two-term fp16 MFMAs (developer builds -DC32_HX=<mask>).
cd "$(dirname "$0")/../.." && ROOT=$(pwd)
mkdir -p gpurun_out/hx
python tools/r06/hx_check.py base 2>&1 | tail -2
ABOPT_LIB_PATH=$ROOT/ab_opt_amd/variants/libabopt_hx1.so ABOPT_DEV_ZTERMS=1 python tools/r06/hx_check.py hx1 base 2>&1 | tail -5
ABL_N=32 ABL_L=256 bash tools/r05/abl.sh hx_abl base hx1,ABOPT_DEV_ZTERMS=1 hx3,ABOPT_DEV_ZTERMS=1 hx7,ABOPT_DEV_ZTERMS=1 2>&1 | tail -12
