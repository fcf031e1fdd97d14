#!/bin/bash
# the refinement in kernel instances of its own: cycles per solve of the default instance, the option's own A/B, the GPU tier
mkdir -p gpurun_out/s29
python tools/phase_profile.py 1024 mpc > gpurun_out/s29/phase_mpc.json 2>/dev/null
python tools/phase_profile.py 1024 > gpurun_out/s29/phase_cold.json 2>/dev/null
python -m pytest tests -q -m gpu > gpurun_out/s29/tests.log 2>&1; tail -n 3 gpurun_out/s29/tests.log
for r in 0 1 0 1; do python bench.py --no-cpu --no-extras --no-parity --refine $r > gpurun_out/s29/head_r${r}_$RANDOM.json 2>/dev/null; done
python bench.py --no-cpu --no-extras --no-parity --refine 1 --tol 1e-6 > gpurun_out/s29/tol6_r1.json 2>/dev/null
python bench.py --no-cpu --no-extras --no-parity --refine 0 --tol 1e-6 > gpurun_out/s29/tol6_r0.json 2>/dev/null
python - <<'P'
import json,glob
d=json.load(open('gpurun_out/s29/phase_mpc.json')); c=d['cycles_per_solve']; print('mpc', {k:int(c[k]) for k in ('total','setup','factor','assemble','resid','linesearch','step','solve','jac')}, d['kernel_ms_p50'])
d=json.load(open('gpurun_out/s29/phase_cold.json')); print('cold', d['total_cycles_per_iter'], d['kernel_ms'])
for f in sorted(glob.glob('gpurun_out/s29/*.json')):
    if 'phase' in f: continue
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value']), d['ms_per_step'], d['mean_iters'], d['solved_fraction'], (d.get('cold_solve') or {}).get('solves_per_s'))
P
