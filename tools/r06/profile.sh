#!/bin/bash
# Round-6 evidence set on ONE box:  gpurun --timeout 2400 -- 'bash tools/r06/profile.sh gpurun_out/r06_j'
# 1. the driver's bench command (full line)   2. rocprofv3 --kernel-trace --stats of the SAME command   3. PMC passes (one group per run,
# --kernel-trace only) on a short eager run, digested for the block's kernels   4. HBM read traffic of the dominant kernel by source (FETCH_SIZE
# passes of developer builds with one stream removed: W_out terms / key-value fragments / z + bias cache)   5. steady-state training step   6. other shapes
cd "$(dirname "$0")/../.." && ROOT=$(pwd) && OUT=$ROOT/${1:-gpurun_out/r06_j} && mkdir -p $OUT && TAG=$(basename $OUT)
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $OUT/bench_full.log 2> $OUT/bench_full.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_profiled.log 2>&1
python $ROOT/tools/rocprof_summary.py $OUT/stats > $OUT/kernel_stats.txt
rm -rf $OUT/stats
# the same with replayed launches only (no eager_events / two-launch passes): the average that roofline.avg_launch_ms must agree with
rocprofv3 --kernel-trace --stats -d $OUT/stats_r -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --replay-only > $OUT/bench_profiled_replay_only.log 2>&1
python $ROOT/tools/rocprof_summary.py $OUT/stats_r > $OUT/kernel_stats_replay_only.txt
rm -rf $OUT/stats_r
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
  { echo "# rocprofv3 --pmc passes (one counter group per run, --kernel-trace only; tools/r06/profile.sh) on: $CMD  (N=32, L=256), MI355X";
    python tools/pmc_digest.py $OUT/pmc --kernel $k $( [ $k = ipa_core ] && echo "--json $OUT/ipa_core_traffic.json --source profiles/${TAG}_pmc_ipa_core.txt" );
    python tools/pmc_summary.py $OUT/pmc --kernel $k; } > $OUT/pmc_$k.txt
done
rm -rf $OUT/pmc
# ---- 4. read traffic by source (needs the developer builds abl1 / abl2 / fabl1 under ab_opt_amd/variants; PROFILE_PARTS=123 stops here)
if [ "${PROFILE_PARTS:-123456}" = 123 ]; then tail -1 $OUT/bench_full.log | cut -c1-1200; head -14 $OUT/kernel_stats.txt | cut -c1-60,92-140; head -8 $OUT/kernel_stats_replay_only.txt | cut -c1-60,92-140; exit 0; fi
cd /tmp
{ echo "# HBM read traffic of the dominant kernel by source: rocprofv3 --pmc FETCH_SIZE (KiB; x2 = the gfx950 correction of MI355X_MICROARCH.md) of developer builds with ONE stream removed";
  echo "# (timing-only builds: their results are wrong by construction).  Command: $CMD  with ABOPT_LIB_PATH=<variant> [ABOPT_FUSE_TAIL=0]"; } > $OUT/traffic_by_source.txt
for spec in "base:" "fabl1:" "base:ABOPT_FUSE_TAIL=0" "abl2:ABOPT_FUSE_TAIL=0" "abl1:ABOPT_FUSE_TAIL=0"; do
  v=${spec%%:*}; e=${spec##*:}
  lib=$ROOT/ab_opt_amd/variants/libabopt_$v.so; [ $v = base ] && lib=$ROOT/ab_opt_amd/libabopt_hip.so
  [ -f $lib ] || { echo "$spec: variant missing" >> $OUT/traffic_by_source.txt; continue; }
  env ABOPT_LIB_PATH=$lib ABOPT_CORE32=1 $e rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/tr_$v --output-format csv -- $CMD > $OUT/tr_$v.log 2>&1
  python - "$OUT/tr_$v" "$spec" >> $OUT/traffic_by_source.txt <<'PY'
import csv, glob, sys, os
tot, n, us = 0.0, 0, 0.0
for f in glob.glob(os.path.join(sys.argv[1], '**', '*_counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ipa_core32' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            tot += float(r['Counter_Value']); n += 1; us += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('%-28s %4d launches  FETCH_SIZE %10.1f KiB -> read %7.1f MB per launch   (%.1f us per launch under the counter pass)' % (sys.argv[2], n, tot / max(n, 1), tot / max(n, 1) * 1024 * 2 / 1e6, us / max(n, 1)))
PY
  rm -rf $OUT/tr_$v
done
cd $ROOT
# ---- 5. training, 6. shapes
bash tools/r05/train_prof.sh $TAG/train > /dev/null 2>&1
bash tools/r03_shapes.sh $TAG/shapes > /dev/null 2>&1
tail -1 $OUT/bench_full.log | cut -c1-1800
head -14 $OUT/kernel_stats.txt | cut -c1-60,92-140
cat $OUT/traffic_by_source.txt
