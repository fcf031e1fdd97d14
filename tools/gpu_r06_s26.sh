#!/bin/bash
# refinement of regularised steps (omgx_options.refine, ABI 9): GPU tests, A/B of the bench line
mkdir -p gpurun_out/s26
python -m pytest tests -x -q -m gpu > gpurun_out/s26/tests.log 2>&1; tail -n 4 gpurun_out/s26/tests.log
for rep in 1 2; do for r in 0 1; do
python bench.py --no-cpu --no-extras --no-parity --refine $r > gpurun_out/s26/head_r${r}_$rep.json 2>/dev/null
python bench.py --no-cpu --no-extras --no-parity --refine $r --streams 1 > gpurun_out/s26/one_r${r}_$rep.json 2>/dev/null
done; done
python bench.py --no-cpu --refine 0 > gpurun_out/s26/full_r0.json 2>/dev/null
python bench.py --no-cpu > gpurun_out/s26/full_r1.json 2> gpurun_out/s26/full_r1.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/s26/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        s=d.get('sustained') or {}
        print(f.split('/')[-1], round(d['value']), 'ms/step %.4f' % d['ms_per_step'], 'iters', d.get('mean_iters'), 'max', d.get('max_iters_in_a_step'), 'cold', (d.get('cold_solve') or {}).get('solves_per_s'), '| sustained', s.get('solves_per_s'), s.get('max_iters'), s.get('solved_fraction'))
        for c in d.get('tolerance_curve', []): print('      ', c.get('settings'), round(c.get('solves_per_s', 0)), (c.get('rollout') or {}).get('solves_per_s'), c.get('solved_fraction'), (c.get('parity') or {}).get('closed_loop_pos_m'))
        if 'parity_at_tol' in d: print('      parity', d['parity_at_tol'].get('closed_loop_pos_m'), d['parity_at_tol'].get('closed_loop_rel'))
    except Exception as e: print(f, 'ERR', e)
P
