#!/usr/bin/env python
"""Condense rocprofv3 output directories (gpurun_out/prof/{stats,fetch,write}) into the small
files committed under profiles/.  Usage: python tools/profile_summary.py <round-tag>"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_COLD = 5      # bench.py --streams 1: four timed cold solves of the side-leg handle + the cold solve of the headline's own handle
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


def issue_passes(tag, out, n_cold=N_COLD):
    """Instruction-fetch / issue decomposition (round 5): every counter of the passes gpurun_out/prof/issue*/, mean per launch
    of ipm_solve_kernel, cold launches (the first `n_cold`: bench.py --streams 1 solves the batch cold five times) and
    receding-horizon steps apart, plus the ratios that decompose the wave-wait share."""
    res = {}
    for d in sorted(glob.glob(os.path.join(PROF, 'issue*'))):
        if not os.path.isdir(d):
            continue
        files = sorted(glob.glob(os.path.join(d, '*', '*_counter_collection.csv')), key=os.path.getmtime)
        if not files:
            continue
        per = {}
        for row in csv.DictReader(open(files[-1])):
            if 'ipm_solve_kernel' in row['Kernel_Name']:
                per.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
        for n, v in per.items():
            res[n] = {'launches': len(v), 'mean_cold': sum(v[:n_cold]) / max(1, len(v[:n_cold])),
                      'mean_warm': sum(v[n_cold:]) / max(1, len(v[n_cold:])), 'pass': os.path.basename(d)}
    def ratio(a, b, k):
        return res[a][k] / max(1.0, res[b][k]) if a in res and b in res else None
    der = {}
    for k in ('mean_cold', 'mean_warm'):
        s_ = k[5:]
        der['wait_any_over_wave_cycles_' + s_] = ratio('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES', k)
        der['wait_inst_any_over_wave_cycles_' + s_] = ratio('SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES', k)
        der['wait_inst_lds_over_wave_cycles_' + s_] = ratio('SQ_WAIT_INST_LDS', 'SQ_WAVE_CYCLES', k)
        der['active_inst_over_wave_cycles_' + s_] = ratio('SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES', k)
        der['salu_share_of_instructions_' + s_] = (res['SQ_INSTS_SALU'][k] / max(1.0, res['SQ_INSTS_SALU'][k] + res['SQ_INSTS_VALU'][k])
                                                    if 'SQ_INSTS_SALU' in res and 'SQ_INSTS_VALU' in res else None)
        der['icache_miss_ratio_' + s_] = ratio('SQC_ICACHE_MISSES', 'SQC_ICACHE_REQ', k)
        der['ifetch_per_wave_cycle_' + s_] = ratio('SQ_IFETCH', 'SQ_WAVE_CYCLES', k)
        der['salu_cycles_over_wave_cycles_' + s_] = ratio('SQ_INST_CYCLES_SALU', 'SQ_WAVE_CYCLES', k)
    res['derived'] = der
    res['note'] = ('separate rocprofv3 --pmc passes (counters only, no trace domain) of `python bench.py --streams 1 --no-cpu '
                   '--no-extras --steps 5 --warmup 1`: launches 0-%d are cold solves of the 1024-agent batch, the rest '
                   'receding-horizon steps; counters the box does not know are absent' % (n_cold - 1))
    json.dump(res, open(os.path.join(out, '%s_pmc_issue.json' % tag), 'w'), indent=1)
    for k, v in der.items():
        print(k, v)
    for k, v in res.items():
        if isinstance(v, dict) and 'mean_warm' in v:
            print('%-28s cold %.4g warm %.4g' % (k, v['mean_cold'], v['mean_warm']))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
    out = os.path.join(ROOT, 'profiles')
    if len(sys.argv) > 2 and sys.argv[2] == 'issue':
        return issue_passes(tag, out)
    stats = newest(os.path.join(PROF, 'stats', '*', '*_kernel_stats.csv'))
    rows = list(csv.reader(open(stats)))
    with open(os.path.join(out, '%s_kernel_stats.csv' % tag), 'w', newline='') as f:
        csv.writer(f).writerows(rows[:12])                      # header + the ten largest kernels
    fetch, write = pmc('FETCH_SIZE'), pmc('WRITE_SIZE')
    hbm = (fetch['mean_kb_per_launch'] + write['mean_kb_per_launch']) * 1024.0
    json.dump({'FETCH_SIZE': fetch, 'WRITE_SIZE': write, 'hbm_bytes_per_launch': hbm, 'cold_launches': N_COLD, 'launches_per_step': 1,
               'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes '
                       '(python bench.py --streams 1 --no-cpu --no-extras --steps 5 --warmup 1: 5 cold solves of the 1024-agent batch + 6 '
                       'receding-horizon steps, one launch per step), kernel ipm_solve_kernel (the instance of the workspace mode, see VGPR / LDS / workgroup columns), 1024 agents; counter unit KB; mean over those launches.'},
              open(os.path.join(out, '%s_pmc_hbm.json' % tag), 'w'), indent=1)
    # further counter passes: mean per launch of every counter, cold launches (the first 4) and warm ones apart
    def multi(sub, names):
        path = newest(os.path.join(PROF, sub, '*', '*_counter_collection.csv'))
        per = {n: [] for n in names}
        for row in csv.DictReader(open(path)):
            if 'ipm_solve_kernel' in row['Kernel_Name'] and row['Counter_Name'] in per:
                per[row['Counter_Name']].append(float(row['Counter_Value']))
        return {n: {'launches': len(v), 'mean_cold': sum(v[:N_COLD]) / max(1, len(v[:N_COLD])),
                    'mean_warm': sum(v[N_COLD:]) / max(1, len(v[N_COLD:]))} for n, v in per.items()}
    mf = multi('mfma', ['SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_F64'])
    for k in ('mean_cold', 'mean_warm'):
        mf['mfma_busy_over_cu_busy_' + k[5:]] = mf['SQ_VALU_MFMA_BUSY_CYCLES'][k] / max(1.0, mf['SQ_BUSY_CU_CYCLES'][k])
    mf['note'] = ('rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 -- python bench.py '
                  '--streams 1 --no-cpu --no-extras --steps 5 --warmup 1; launches 0-4 are the cold solves of the batch, the rest '
                  'receding-horizon steps; 1024 agents')
    json.dump(mf, open(os.path.join(out, '%s_pmc_mfma.json' % tag), 'w'), indent=1)
    lw = multi('lds', ['SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_INSTS_LDS'])
    lw.update(multi('wait', ['SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES']))
    for k in ('mean_cold', 'mean_warm'):
        lw['wait_any_over_wave_cycles_' + k[5:]] = lw['SQ_WAIT_ANY'][k] / max(1.0, lw['SQ_WAVE_CYCLES'][k])
        lw['active_inst_over_wave_cycles_' + k[5:]] = lw['SQ_ACTIVE_INST_ANY'][k] / max(1.0, lw['SQ_WAVE_CYCLES'][k])
        lw['bank_conflict_over_lds_active_' + k[5:]] = lw['SQ_LDS_BANK_CONFLICT'][k] / max(1.0, lw['SQ_LDS_IDX_ACTIVE'][k])
    lw['note'] = 'two separate rocprofv3 --pmc passes (LDS counters; wave wait / issue split), same command as the MFMA pass'
    json.dump(lw, open(os.path.join(out, '%s_pmc_lds_wait.json' % tag), 'w'), indent=1)
    try:
        oc = multi('occ', ['GRBM_GUI_ACTIVE', 'SQ_WAVES', 'SQ_BUSY_CYCLES'])
        oc['note'] = ('rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES, same command: SQ_WAVES = waves launched per kernel '
                      '(persistent workgroups: grid x waves per workgroup); the launch geometry (workgroup size, LDS, VGPR) is in '
                      'the pmc_hbm file')
        json.dump(oc, open(os.path.join(out, '%s_pmc_occupancy.json' % tag), 'w'), indent=1)
    except SystemExit:
        pass
    print('kernel stats:', rows[1][0][:40], rows[1][1:4])
    print('HBM bytes / launch:', hbm)
    print('MFMA busy / CU busy: cold %.3f warm %.3f' % (mf['mfma_busy_over_cu_busy_cold'], mf['mfma_busy_over_cu_busy_warm']))
    print('wait / wave cycles: cold %.3f warm %.3f' % (lw['wait_any_over_wave_cycles_cold'], lw['wait_any_over_wave_cycles_warm']))


if __name__ == '__main__':
    main()
