#!/usr/bin/env python
"""Condense rocprofv3 output directories (gpurun_out/prof/{stats,fetch,write}) into the small
files committed under profiles/.  Usage: python tools/profile_summary.py <round-tag>"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'gpurun_out', 'prof')


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    if not files:
        raise SystemExit('no file matches %s' % pattern)
    return files[-1]


def pmc(name):
    path = newest(os.path.join(PROF, name.split('_')[0].lower(), '*', '*_counter_collection.csv'))
    vals, meta = [], {}
    for row in csv.DictReader(open(path)):
        if 'ipm_solve_kernel' in row['Kernel_Name'] and row['Counter_Name'] == name:
            vals.append(float(row['Counter_Value']))
            meta = {'lds_block_size': row['LDS_Block_Size'], 'vgpr': row['VGPR_Count'],
                    'accum_vgpr': row['Accum_VGPR_Count'], 'sgpr': row['SGPR_Count'],
                    'workgroup': row['Workgroup_Size'], 'grid': row['Grid_Size']}
    return dict(meta, launches=len(vals), mean_kb_per_launch=sum(vals) / max(1, len(vals)),
                min=min(vals), max=max(vals), per_launch_kb=vals)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
    out = os.path.join(ROOT, 'profiles')
    stats = newest(os.path.join(PROF, 'stats', '*', '*_kernel_stats.csv'))
    rows = list(csv.reader(open(stats)))
    with open(os.path.join(out, '%s_kernel_stats.csv' % tag), 'w', newline='') as f:
        csv.writer(f).writerows(rows[:12])                      # header + the ten largest kernels
    fetch, write = pmc('FETCH_SIZE'), pmc('WRITE_SIZE')
    hbm = (fetch['mean_kb_per_launch'] + write['mean_kb_per_launch']) * 1024.0
    json.dump({'FETCH_SIZE': fetch, 'WRITE_SIZE': write, 'hbm_bytes_per_launch': hbm,
               'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes '
                       '(python bench.py --steps 5 --warmup 1 --no-cpu: 3 cold solves + 6 receding-horizon '
                       'steps), kernel ipm_solve_kernel<0>, 1024 agents; counter unit KB; mean over those launches.'},
              open(os.path.join(out, '%s_pmc_hbm.json' % tag), 'w'), indent=1)
    print('kernel stats:', rows[1][0][:40], rows[1][1:4])
    print('HBM bytes / launch:', hbm)


if __name__ == '__main__':
    main()
