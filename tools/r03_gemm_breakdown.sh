#!/bin/bash
# Training step: the launches of abopt::gemm_batched_kernel / slab_sum / colsum (or, with a second argument 'all', of every kernel)
# grouped by grid size (which products cost what)
cd "$(dirname "$0")/.." && ROOT=$(pwd) && OUT=$ROOT/gpurun_out/${1:-gemmbd} && mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python $ROOT/tools/prof_train_full.py > /dev/null 2>&1
python - "$OUT" "${2:-gemm}" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/tr/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if sys.argv[2] == 'all' or 'gemm_batched' in n or 'slab_sum' in n or 'colsum' in n or 'Cijk' in n:
        key = (n.split('(')[0][:48], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
        a = agg[key]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open(out + '/gemm_breakdown.txt', 'w') as fo:
    for k, (c, t) in rows[:70]:
        fo.write('%-50s grid %-7s %-5s %-5s  x%-4d total %9.1f us  avg %8.1f us\n' % (k[0], k[1], k[2], k[3], c, t, t / c))
print(open(out + '/gemm_breakdown.txt').read())
PY
rm -rf $OUT/tr
