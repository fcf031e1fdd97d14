#!/bin/bash
# experiment: row scales known ahead of the Jacobian (upper bound of what carrying the scaling across steps saves)
mkdir -p gpurun_out/s19
for rep in 1 2; do
  for v in base carry; do
    if [ $v = carry ]; then export OMGX_LIB=$PWD/tools/scratch/libomgx_carry.so; else unset OMGX_LIB; fi
    python bench.py --no-parity --no-cpu --no-extras --steps 20 --warmup 5 > gpurun_out/s19/bench_${v}_$rep.json 2> gpurun_out/s19/bench_${v}_$rep.err
    python bench.py --no-parity --no-cpu --no-extras --steps 20 --warmup 5 --streams 1 > gpurun_out/s19/bench1_${v}_$rep.json 2> gpurun_out/s19/bench1_${v}_$rep.err
  done
done
unset OMGX_LIB
python tools/phase_profile.py 1024 mpc > gpurun_out/s19/phase_base.json 2>/dev/null
OMGX_PROF_LIB=$PWD/tools/scratch/libomgx_carry_prof.so python tools/phase_profile.py 1024 mpc > gpurun_out/s19/phase_carry.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/s19/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), d['ms_per_step'], d.get('mean_iters'))
    except Exception as e: print(f, 'ERR', e)
for f in ('base','carry'):
    d=json.load(open('gpurun_out/s19/phase_%s.json'%f)); c=d['cycles_per_solve']
    print(f, d['kernel_ms_p50'], d['iters_per_solve'], {k:int(c[k]) for k in ('total','setup','s_params','s_jac0','s_class','s_init')})
P
