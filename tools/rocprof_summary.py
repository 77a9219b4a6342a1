#!/usr/bin/env python
"""Turn a rocprofv3 `--kernel-trace --stats` result database (rocpd sqlite) into a small text table.

    python tools/rocprof_summary.py gpurun_out/prof1 > profiles/rNN_<what>.txt
"""
import glob
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.match(r'(?:void )?([\w:]+(?:<[^(]{0,60}?>)?)\(', name)
    out = m.group(1) if m else name
    return out if len(out) <= 90 else out[:87] + '...'


def main(path):
    dbs = [path] if path.endswith('.db') else sorted(glob.glob(os.path.join(path, '**', '*_results.db'), recursive=True))
    for db in dbs:
        con = sqlite3.connect(db)
        rows = list(con.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
        print(f'# {os.path.basename(db)}  (durations in microseconds, from rocprofv3 --kernel-trace --stats)')
        print(f'{"kernel":<92}{"calls":>7}{"total_us":>14}{"avg_us":>12}{"pct":>8}')
        total = sum(r[2] for r in rows)
        own = sum(r[2] for r in rows if 'abopt::' in r[0])
        print(f'# GPU time in kernels: {total / 1e3:.1f} us total, {own / 1e3:.1f} us ({100 * own / max(total, 1):.1f} %) in abopt:: kernels; {len(rows)} distinct kernels')
        for n, c, tot, avg, pct in (rows if '--all' in sys.argv else rows[:25])[:60]:
            print(f'{short(n):<92}{c:>7}{tot / 1e3 if False else tot:>14.1f}{avg:>12.2f}{pct:>8.2f}')
        print()


if __name__ == '__main__':
    main([a for a in sys.argv[1:] if not a.startswith('--')][0])
