#!/bin/bash
# the whole-manoeuvre leg and the one-launch form on other seeds of the workload (the committed bundle's generator with another seed)
python - <<'P' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_sustained_seeds.txt
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'omg-tools_amd')
import bench
from omgtools import workloads
dev = torch.device('cuda', 0)
opts = dict(tol=1e-3, max_iter=300)
print('whole manoeuvre (bench.py sustained leg: 120 updates, stop rule on) and the same as one rollout launch, 1024 agents, tol 1e-3; seed None = the bench workload')
for seed in (None, 11, 12, 13, 14):
    problem, P = workloads.holonomic_p2p(1024) if seed is None else workloads.holonomic_p2p(1024, seed=seed)
    if seed is None:
        bench.sustained_leg(problem, P, opts, 120, dev)      # (lazy loading)
    r = bench.sustained_leg(problem, P, opts, 120, dev)
    q = bench.manoeuvre_rollout(problem, P, opts, 120, dev)
    print('seed %-5s per-step path %9.0f solves/s  %.4f ms/update  %d solves  %.4f iterations/solve  slowest %3d  solved %.6f  arrived %.3f (p50 update %d, last %d) | one launch %9.0f solves/s  slowest %3d  solved %.6f' % (
        seed, r['solves_per_s'], r['ms_per_update'], r['solves'], r['mean_iters'], r['max_iters'], r['solved_fraction'], r['arrived_fraction'], r['updates_to_arrival_p50'], r['updates_to_arrival_max'],
        q['solves_per_s'], q['max_iters'], q['solved_fraction']), flush=True)
P
