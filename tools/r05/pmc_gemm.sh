#!/bin/bash
# PMC passes over one GEMM shape: bash tools/r05/pmc_gemm.sh <outdir> B M N K at bt br
cd "$(dirname "$0")/../.." && ROOT=$(pwd); OUT=$ROOT/${1:-gpurun_out/pmc_gemm}; shift; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
CMD="python $ROOT/tools/r05/gemm_one.py $* 20"
$CMD 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats -d $OUT/st --output-format csv -- $CMD > $OUT/st.log 2>&1
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i --output-format csv -- $CMD > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done <<'GRPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
GRPS
cd $ROOT && python tools/pmc_summary.py $OUT --kernel gemm_batched_kernel 2>&1 | tail -40
python tools/rocprof_summary.py $OUT/st 2>/dev/null | head -8
