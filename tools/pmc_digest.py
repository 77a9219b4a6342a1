#!/usr/bin/env python
"""Header lines for a profiles/rNN_x_pmc_<kernel>.txt file and (for the IPA core) the record of profiles/ipa_core_traffic.json.
    python tools/pmc_digest.py <pmc dir> --kernel ipa_core [--json out.json --source 'profiles/...'] [--N 32 --L 256]
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE come from SEPARATE --pmc passes, are in KiB, and FETCH_SIZE
counts 64 B per 128-B request of a wide coalesced read on gfx950, so it is doubled."""
import collections
import csv
import glob
import json
import os
import sys


def arg(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def main():
    d, kern = sys.argv[1], arg('--kernel')
    acc, dur = collections.defaultdict(lambda: [0.0, 0]), [0.0, 0]
    seen = set()
    for f in sorted(glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True)):
        for r in csv.DictReader(open(f)):
            if kern not in r['Kernel_Name']:
                continue
            acc[r['Counter_Name']][0] += float(r['Counter_Value']); acc[r['Counter_Name']][1] += 1
            key = (f, r['Dispatch_Id'])
            if key not in seen:
                seen.add(key)
                dur[0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; dur[1] += 1
    c = {k: v[0] / v[1] for k, v in acc.items()}
    us = dur[0] / max(dur[1], 1)
    print(f'# kernel filter "{kern}": {dur[1]} profiled dispatches over all passes, average {us:.1f} us under the counter passes')
    rec = None
    if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        rd, wr = c['FETCH_SIZE'] * 1024 * 2, c['WRITE_SIZE'] * 1024
        print(f'#   read  = {c["FETCH_SIZE"]:.1f} KiB * 1024 * 2 (gfx950 correction) = {rd / 1e6:.1f} MB / launch;  written = {c["WRITE_SIZE"]:.1f} KiB * 1024 = {wr / 1e6:.1f} MB / launch'
              f'  => {(rd + wr) / 1e6:.1f} MB HBM traffic per launch')
        rec = dict(bytes_per_launch=int(rd + wr))
    if 'SQ_LDS_BANK_CONFLICT' in c and c.get('SQ_LDS_IDX_ACTIVE'):
        print(f'#   LDS bank-conflict cycles / LDS active = {c["SQ_LDS_BANK_CONFLICT"]:.0f} / {c["SQ_LDS_IDX_ACTIVE"]:.0f} = {100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]:.1f} %')
    if 'TCC_HIT_sum' in c:
        print(f'#   L2 hit rate = {100 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]):.1f} %')
    if 'TCP_TCC_READ_REQ_sum' in c and c.get('TCP_TCC_READ_REQ_LATENCY_sum'):
        lat = c['TCP_TCC_READ_REQ_LATENCY_sum'] / c['TCP_TCC_READ_REQ_sum']
        print(f'#   L1 -> L2 read requests {c["TCP_TCC_READ_REQ_sum"] / 1e6:.2f} M per launch, average latency {lat:.0f} clk')
    if 'SQ_INSTS_MFMA' in c:
        print(f'#   wave instructions per launch: MFMA {c["SQ_INSTS_MFMA"] / 1e6:.2f} M, VALU {c.get("SQ_INSTS_VALU", 0) / 1e6:.2f} M, LDS {c.get("SQ_INSTS_LDS", 0) / 1e6:.2f} M, VMEM {c.get("SQ_INSTS_VMEM", 0) / 1e6:.2f} M')
    if arg('--json') and rec:
        rec.update(N=int(arg('--N', 32)), L=int(arg('--L', 256)),
                   source=f'{arg("--source", d)}: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, separate passes, avg of {acc["FETCH_SIZE"][1]} launches')
        with open(arg('--json'), 'w') as fh:
            json.dump([rec], fh, indent=1)


main()
