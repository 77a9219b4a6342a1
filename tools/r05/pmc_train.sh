#!/bin/bash
# PMC passes over two training steps (config 5), digested for a few kernels: bash tools/r05/pmc_train.sh <outdir> kernel_substring [kernel_substring ...]
cd "$(dirname "$0")/../.." && ROOT=$(pwd); OUT=$ROOT/${1:-gpurun_out/pmc_train}; shift; mkdir -p $OUT
export TMPDIR=/tmp STEPS=2; cd /tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i --output-format csv -- python $ROOT/tools/prof_train_full.py > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done <<'GRPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
GRPS
cd $ROOT
for k in "$@"; do echo "==== $k"; python tools/pmc_summary.py $OUT --kernel $k 2>&1 | grep -v "^#" ; done
rm -rf $OUT/g[0-9]*/
