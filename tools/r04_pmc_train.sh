#!/bin/bash
# PMC passes over the training step for chosen kernels:  bash tools/r04_pmc_train.sh <tag> <kernel substring> [...]
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-pmc_train} && mkdir -p $OUT
shift
export TMPDIR=/tmp
cd /tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc/g$i --output-format csv -- python $ROOT/tools/prof_train_full.py > $OUT/pmc_g$i.log 2>&1 || echo "group $i failed"
done <<'GRPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
GRPS
cd $ROOT
for k in "$@"; do python tools/pmc_digest.py $OUT/pmc --kernel $k > $OUT/pmc_$k.txt; python tools/pmc_summary.py $OUT/pmc --kernel $k >> $OUT/pmc_$k.txt; cat $OUT/pmc_$k.txt | cut -c1-170; done
rm -rf $OUT/pmc
