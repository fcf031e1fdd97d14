#!/bin/bash
# round 6, step 5: setup fusion (scaled Jacobian written by the row owners, one start-value pass / reduction) -- GPU tier, bit identity against
# the previous library over a receding-horizon loop, phase cycles and bench A/B (previous library in tools/scratch/)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s8
( timeout 1500 python -m pytest tests/test_gpu_prepare.py tests/test_gpu_rollout.py tests/test_gpu_glue.py tests/test_gpu_solver.py tests/test_gpu_admm.py tests/test_gpu_batch_mpc.py -q > gpurun_out/s8/gputests.log 2>&1; echo "rc $?" >> gpurun_out/s8/gputests.log )
tail -6 gpurun_out/s8/gputests.log
cat > /tmp/bits.py <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'omg-tools_amd'))
import numpy as np, torch
from omgtools import workloads
from omgtools.batch import BatchP2P
out = {}
for name, B, steps in (('holonomic_p2p', 256, 12), ('quadrotor_p2p', 32, 4), ('holonomic3d_p2p', 16, 3)):
    problem, P = getattr(workloads, name)(B)
    m = BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=dict(P.get('solver_options') or {}, tol=1e-3, max_iter=300))
    m.solve_cold()
    for _ in range(steps): m.step()
    out[name + '_x'], out[name + '_lam'], out[name + '_it'] = m.host('x'), m.host('lam'), m.host('iters')
    m.solver.close()
np.savez(sys.argv[1], **out)
PY
OMGX_LIB=$GRAFT_REPO_ROOT/tools/scratch/libomgx_prev.so python /tmp/bits.py /tmp/bits_prev.npz
python /tmp/bits.py /tmp/bits_new.npz
python - <<'PY'
import numpy as np
a, b = np.load('/tmp/bits_prev.npz'), np.load('/tmp/bits_new.npz')
print('bit identity new vs previous library:', {k: bool(np.array_equal(a[k], b[k])) for k in a.files})
PY
for rep in 1 2; do
  for lib in prev new; do
    if [ $lib = prev ]; then export OMGX_LIB=$GRAFT_REPO_ROOT/tools/scratch/libomgx_prev.so; else unset OMGX_LIB; fi
    timeout 300 python bench.py --no-cpu --no-extras > gpurun_out/s8/bench_${lib}_r${rep}.json 2> gpurun_out/s8/bench_${lib}_r${rep}.err
    timeout 300 python bench.py --streams 1 --no-cpu --no-extras > gpurun_out/s8/bench1_${lib}_r${rep}.json 2> gpurun_out/s8/bench1_${lib}_r${rep}.err
  done
done
unset OMGX_LIB
OMGX_PREPARE=0 OMGX_PROF_LIB=$GRAFT_REPO_ROOT/tools/scratch/libomgx_prof_prev.so python tools/phase_profile.py 1024 mpc > gpurun_out/s8/phase_mpc_prev.json 2> gpurun_out/s8/phase_prev.err
python tools/phase_profile.py 1024 mpc > gpurun_out/s8/phase_mpc_new.json 2> gpurun_out/s8/phase_new.err
python tools/phase_profile.py 1024 > gpurun_out/s8/phase_cold_new.json 2>> gpurun_out/s8/phase_new.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/s8/bench*.json')):
    try:
        d = json.load(open(f))
        print(f, 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'k_ms %.3f' % d['roofline']['kernel_ms'], 'cold %.0f' % d['cold_solve']['solves_per_s'], 'iters %.3f' % d['mean_iters'])
    except Exception as e:
        print(f, 'ERR', e)
for w in ('prev', 'new'):
    d = json.load(open('gpurun_out/s8/phase_mpc_%s.json' % w)); c = d['cycles_per_solve']
    print(w, 'total %.0f setup %.0f' % (c['total'], c['setup']), {k: round(c[k]) for k in ('s_params', 's_jac0', 's_class', 's_init')})
PY
