#!/usr/bin/env python
"""Steady-state kernel statistics of ONE training step: the difference of two rocprofv3 runs of tools/prof_train_full.py with different step counts
(the first steps allocate the optimizer state -- 2 x 207 zero fills -- and pack weights; profiling the first four steps, as rounds 3-4 did, books
those one-offs as per-step cost).    python tools/r05/train_steady.py <dir with steps A> <dir with steps B> <A> <B>"""
import glob, os, re, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rocprof_summary import short


def load(path):
    db = sorted(glob.glob(os.path.join(path, '**', '*_results.db'), recursive=True))[0]
    con = sqlite3.connect(db)
    return {r[0]: (r[1], r[2]) for r in con.execute('select name, total_calls, total_duration from top_kernels')}


a, b, na, nb = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    dc, dt = (cb - ca) / (nb - na), (tb - ta) / (nb - na)
    if dc > 0:          # a kernel with the same call count in both runs is a one-off of the process (its duration jitter is not per-step time)
        rows.append((dt, dc, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
own = sum(r[0] for r in rows if 'abopt::' in r[2])
print(f'# steady-state training step (config 5: N=16, L=256, AbDesign flavour, FusedAdam): per-step kernel time = (run with {nb} steps - run with {na} steps) / {nb - na}, rocprofv3 --kernel-trace --stats')
print(f'# GPU time per step {tot / 1e3:.3f} ms, {own / 1e3:.3f} ms ({100 * own / tot:.1f} %) in abopt:: kernels; {sum(r[1] for r in rows):.0f} launches per step, {sum(r[1] for r in rows if "abopt::" not in r[2]):.0f} of them outside abopt::')
print(f'{"kernel":<92}{"calls/step":>11}{"us/step":>12}{"avg_us":>10}{"pct":>7}')
for dt, dc, k in rows[:70]:
    print(f'{short(k):<92}{dc:>11.2f}{dt:>12.1f}{dt / dc:>10.2f}{100 * dt / tot:>7.2f}')
