#!/bin/bash
# Developer tool: rocprofv3 PMC passes (one counter group per run, --kernel-trace only) over the IPA-core microbench.
#   gpurun -- 'bash tools/pmc_ipa.sh gpurun_out/pmc_x'
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/pmc}; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python tools/bench_ipa_cached.py ${2:-32} ${3:-256} 3"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i --output-format csv -- $CMD > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done <<'GRPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_TA_BUSY_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
GRPS
python tools/pmc_summary.py $OUT --kernel ipa_core_kernel
