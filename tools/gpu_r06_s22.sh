#!/bin/bash
# hardware-queue placement of the sub-batch streams: what a process creates first decides whether the three launches of a step overlap
mkdir -p gpurun_out/s22
cat > /tmp/q.py <<'P'
import sys, os, time, torch
sys.path.insert(0, 'omg-tools_amd')
from omgtools import workloads
from omgtools.batch import receding_horizon_batch, BatchP2P
from omgtools.backend import BatchSolver
dev = torch.device('cuda', 0)
problem, P = workloads.holonomic_p2p(1024)
opts = dict(tol=1e-3, max_iter=300)
mode = sys.argv[1]
if mode in ('handle_first', 'two_handles_first'):
    dummy = BatchSolver(problem.father.template, 8)
if mode == 'two_handles_first':
    dummy2 = BatchP2P(problem, dict(P, p=P['p'][:64], x0=P['x0'][:64]), ops='hip', device=dev, options=opts)
    dummy2.solve_cold(bends=()); dummy2.step()
if mode == 'other_streams_first':
    ss = [torch.cuda.Stream() for _ in range(5)]
    for q in ss:
        with torch.cuda.stream(q): torch.zeros(4, device=dev).add_(1)
for rep in range(3):
    rh = receding_horizon_batch(problem, P, device=dev, n_streams=3, options=opts)
    rh.solve_cold(bends=())
    for _ in range(5): rh.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): rh.step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, 'prio', os.environ.get('OMGX_STREAM_PRIORITY', '-1'), 'queues', os.environ.get('GPU_MAX_HW_QUEUES', '-'), 'instance', rep, 'solves/s %.0f' % (1024 * 20 / dt), flush=True)
    rh.close()
P
for mode in streamed_first handle_first two_handles_first other_streams_first; do
  python /tmp/q.py $mode 2>/dev/null | grep solves | grep -v "instance 0"
  OMGX_STREAM_PRIORITY=0 python /tmp/q.py $mode 2>/dev/null | grep solves | grep -v "instance 0"
done
GPU_MAX_HW_QUEUES=8 python /tmp/q.py handle_first 2>/dev/null | grep solves | grep -v "instance 0"
GPU_MAX_HW_QUEUES=8 python /tmp/q.py two_handles_first 2>/dev/null | grep solves | grep -v "instance 0"
