#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s11
for rep in 1 2; do for l in base pf; do
OMGX_PROF_LIB=$GRAFT_REPO_ROOT/tools/scratch/libomgx_prof_$l.so python tools/phase_profile.py 1024 mpc > gpurun_out/s11/phase_${l}_r$rep.json 2> gpurun_out/s11/phase_${l}_r$rep.err
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/s11/phase_*.json')):
    c = json.load(open(f))['cycles_per_solve']
    print(f, {k: round(c[k]) for k in ('total', 'assemble', 'a_zero', 'a_pairs', 'a_tcol', 'a_hess', 'a_rest')})
PY
