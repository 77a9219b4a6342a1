#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per kernel from *_counter_collection.csv files.
    python tools/pmc_summary.py gpurun_out/pmc_sq gpurun_out/pmc_fetch [--kernel ipa_core]"""
import csv, glob, os, sys, collections

def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    kern = None
    if '--kernel' in sys.argv:
        kern = sys.argv[sys.argv.index('--kernel') + 1]
        args = [a for a in args if a != kern]
    for d in args:
        for f in sorted(glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True)):
            acc = collections.defaultdict(lambda: [0.0, 0])
            dur = collections.defaultdict(lambda: [0.0, 0])
            seen = set()
            for r in csv.DictReader(open(f)):
                name = r['Kernel_Name'].split('(')[0]
                if kern and kern not in name:
                    continue
                k = (name, r['Counter_Name'])
                acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
                if (r['Dispatch_Id'], name) not in seen:
                    seen.add((r['Dispatch_Id'], name))
                    dur[name][0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; dur[name][1] += 1
                    meta = (r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'], r['Grid_Size'], r['Workgroup_Size'])
                    dur[name].append(meta) if len(dur[name]) == 2 else None
            print(f'# {f}')
            for name, v in dur.items():
                print(f'{name}: {v[1]} dispatches, avg {v[0]/v[1]:.1f} us; vgpr/agpr/sgpr/lds/scratch/grid/wg = {v[2] if len(v)>2 else ""}')
            for (name, c), (tot, n) in sorted(acc.items()):
                print(f'  {name[:60]:<60} {c:<28} avg/dispatch {tot/n:>18.1f}')

main()
