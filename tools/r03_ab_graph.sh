#!/bin/bash
# Round 3, same-box A/B of the launch modes of the sampling loop: eager with / without the per-launch HIP events, hipGraph replay
# without events, hipGraph with the event records captured inside.  Usage (GPU box): bash tools/r03_ab_graph.sh
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
C="--no-cpu-baseline --no-secondary --repeats 5"
for v in "--graph off" "--graph off --no-prof" "--graph on" "--graph on --graph-events" "--graph off" "--graph on"; do
  echo "=== bench.py $v" >> gpurun_out/ab_graph.log
  timeout 600 python bench.py $C $v >> gpurun_out/ab_graph.log 2>> gpurun_out/ab_graph.err
done
grep -E "^===|^\{" gpurun_out/ab_graph.log | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('==='): print(l.strip()); continue
    d = json.loads(l); r = d['roofline']
    print('   value %.0f (min %.0f max %.0f)  ms/step %.4f  core %.4f ms frac %.3f  instr %s' % (d['value'], d['value_min'], d['value_max'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['instrumented_ms_per_step']))
" | tee gpurun_out/ab_graph_summary.txt
