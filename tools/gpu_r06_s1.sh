#!/bin/bash
# round 6, step 1: the setup kernel (ipm_prepare_kernel) -- bit identity tests, then A/B of the bench with it on / off in one call
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
( timeout 900 python -m pytest tests/test_gpu_prepare.py tests/test_gpu_rollout.py tests/test_gpu_solver.py tests/test_gpu_batch_mpc.py -q -x > gpurun_out/s1/tests.log 2>&1; echo "rc $?" >> gpurun_out/s1/tests.log )
tail -15 gpurun_out/s1/tests.log
for rep in 1 2; do
  for on in 1 0; do
    OMGX_PREPARE=$on timeout 300 python bench.py --no-cpu --no-extras > gpurun_out/s1/bench_p${on}_r${rep}.json 2> gpurun_out/s1/bench_p${on}_r${rep}.err
    OMGX_PREPARE=$on timeout 300 python bench.py --streams 1 --no-cpu --no-extras > gpurun_out/s1/bench1_p${on}_r${rep}.json 2> gpurun_out/s1/bench1_p${on}_r${rep}.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/s1/bench*.json')):
    try:
        d = json.load(open(f))
        print(f, 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'k_ms %.3f' % d['roofline']['kernel_ms'], 'cold %.0f' % d['cold_solve']['solves_per_s'], 'iters %.3f' % d['mean_iters'], 'p50 %.3f' % d['p50_batch_latency_ms'])
    except Exception as e:
        print(f, 'ERR', e)
PY
