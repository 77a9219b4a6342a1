#!/bin/bash
# Collect the evidence behind one round's numbers (run on the GPU box):
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh gpurun_out/r03_a'
# 1. full default bench line   2. rocprofv3 --kernel-trace --stats of the same workload   3. PMC passes (one group per run,
# --kernel-trace only) on a short eager run, digested for the three kernels of a GABlock (tools/pmc_digest.py, tools/pmc_summary.py).
cd "$(dirname "$0")/.."
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/prof}; mkdir -p $OUT
TAG=$(basename $OUT)
export TMPDIR=/tmp
python bench.py > $OUT/bench_full.log 2> $OUT/bench_full.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python $ROOT/bench.py --no-cpu-baseline --no-secondary > $OUT/bench_profiled.log 2>&1
python $ROOT/tools/rocprof_summary.py $OUT/stats > $OUT/kernel_stats.txt
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
for k in ipa_core node_frags_kernel out_ln_mlp_kernel; do
  { echo "# rocprofv3 --pmc passes (one counter group per run, --kernel-trace only; tools/profile_round.sh) on: $CMD  (N=32, L=256), MI355X";
    python tools/pmc_digest.py $OUT/pmc --kernel $k $( [ $k = ipa_core ] && echo "--json $OUT/ipa_core_traffic.json --source profiles/${TAG}_pmc_ipa_core.txt" );
    python tools/pmc_summary.py $OUT/pmc --kernel $k; } > $OUT/pmc_$k.txt
done
rm -rf $OUT/stats $OUT/pmc
# 4. the training step (config 5) and the shapes other than the bench's
bash tools/r03_train_prof.sh $TAG/train > /dev/null 2>&1
bash tools/r03_shapes.sh $TAG/shapes > /dev/null 2>&1
tail -1 $OUT/bench_full.log | cut -c1-600
head -14 $OUT/kernel_stats.txt | cut -c1-60,92-140
