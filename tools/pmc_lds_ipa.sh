#!/bin/bash
# Developer tool: LDS / issue counters of the IPA core on the micro-benchmark (one rocprofv3 --pmc pass).
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/pmc_lds}; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/g1 --output-format csv -- python tools/bench_ipa_cached.py 32 256 3 > $OUT/g1.log 2>&1
python tools/pmc_summary.py $OUT --kernel ipa_core_kernel | grep -E "dispatches|LDS|MFMA_BUSY"
tail -1 $OUT/g1.log
rm -rf $OUT/g1
