#!/bin/bash
# Round-4 evidence set on ONE box:  gpurun --timeout 1800 -- 'bash tools/r04_profile.sh gpurun_out/r04_a'
# 1. the driver's bench command (full line)   2. rocprofv3 --kernel-trace --stats of the SAME command   3. PMC passes (one group per run,
# --kernel-trace only) on a short eager run, digested for the block's kernels   4. training step + other shapes
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/r04_a}; mkdir -p $OUT
TAG=$(basename $OUT)
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $OUT/bench_full.log 2> $OUT/bench_full.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_profiled.log 2>&1
python $ROOT/tools/rocprof_summary.py $OUT/stats > $OUT/kernel_stats.txt
rm -rf $OUT/stats
CMD="python $ROOT/bench.py --steps 4 --warmup 1 --repeats 1 --graph off --no-prof --no-cpu-baseline --no-secondary"
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc/g$i --output-format csv -- $CMD > $OUT/pmc_g$i.log 2>&1 || echo "group $i failed: $grp"
done <<'GRPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
FETCH_SIZE GRBM_GUI_ACTIVE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
GRPS
cd $ROOT
for k in ipa_core node_frags_kernel; do
  { echo "# rocprofv3 --pmc passes (one counter group per run, --kernel-trace only; tools/r04_profile.sh) on: $CMD  (N=32, L=256), MI355X";
    python tools/pmc_digest.py $OUT/pmc --kernel $k $( [ $k = ipa_core ] && echo "--json $OUT/ipa_core_traffic.json --source profiles/${TAG}_pmc_ipa_core.txt" );
    python tools/pmc_summary.py $OUT/pmc --kernel $k; } > $OUT/pmc_$k.txt
done
rm -rf $OUT/pmc
tail -1 $OUT/bench_full.log | cut -c1-1500
head -14 $OUT/kernel_stats.txt | cut -c1-60,92-140
