#!/bin/bash
# whole-manoeuvre leg with three and four sub-batches
python - <<'P' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s33.txt
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'omg-tools_amd')
import bench
import omgtools.batch as ob
from omgtools import workloads
dev = torch.device('cuda', 0)
problem, P = workloads.holonomic_p2p(1024)
opts = dict(tol=1e-3, max_iter=300)
for ns in (3, 4, 3, 4, 2):
    ob.product_path_streams = lambda *a, **k: ns
    r = bench.sustained_leg(problem, P, opts, 120, dev)
    print('sub-batches', ns, 'solves/s %.0f' % r['solves_per_s'], 'ms/update %.4f' % r['ms_per_update'], flush=True)
P
