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
        for n, c, tot, avg, pct in rows[:25]:
            print(f'{short(n):<92}{c:>7}{tot:>14.1f}{avg:>12.2f}{pct:>8.2f}')
        print()


if __name__ == '__main__':
    main(sys.argv[1])
