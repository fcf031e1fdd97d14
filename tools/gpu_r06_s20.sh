#!/bin/bash
# stop rule of the solve kernel (ABI 9): tests, same-box A/B of the headline against the previous library, the bench line with the new sustained leg
mkdir -p gpurun_out/s20
python -m pytest tests/test_gpu_batch_mpc.py tests/test_gpu_prepare.py tests/test_gpu_rollout.py -x -q -m gpu > gpurun_out/s20/tests.log 2>&1; tail -n 3 gpurun_out/s20/tests.log
for rep in 1 2; do
  for v in v8 v9; do
    if [ $v = v8 ]; then export OMGX_LIB=$PWD/tools/scratch/libomgx_v8.so; else unset OMGX_LIB; fi
    python - > gpurun_out/s20/ab_${v}_$rep.txt 2>&1 <<'P'
import sys, time, torch, numpy as np
sys.path.insert(0, 'omg-tools_amd')
from omgtools import workloads
from omgtools.batch import receding_horizon_batch, BatchP2P
dev = torch.device('cuda', 0)
problem, P = workloads.holonomic_p2p(1024)
for streams in (3, 1):
    rh = receding_horizon_batch(problem, P, device=dev, n_streams=streams, options=dict(tol=1e-3, max_iter=300)) if streams > 1 else BatchP2P(problem, P, ops='hip', device=dev, options=dict(tol=1e-3, max_iter=300))
    rh.solve_cold(bends=())
    for _ in range(5): rh.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): rh.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('streams', streams, 'solves/s %.0f' % (1024 * 20 / dt), 'iters %.4f' % float(rh.iters.double().mean()))
    (rh.close() if streams > 1 else rh.solver.close())
P
  done
done
unset OMGX_LIB
cat gpurun_out/s20/ab_*.txt
python bench.py > gpurun_out/s20/bench.json 2> gpurun_out/s20/bench.err; tail -n 2 gpurun_out/s20/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/s20/bench.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step']); s=d['sustained']; print({k:v for k,v in s.items() if k not in ('windows','note')}); [print(w) for w in s.get('windows',[])]
P
