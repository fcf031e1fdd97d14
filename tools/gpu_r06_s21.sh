#!/bin/bash
# why is a second whole-manoeuvre leg in the same process slower?  order / stream experiments
mkdir -p gpurun_out/s21
for order in "0 0 1 1" "1 1 0 0" "0 1 0 1"; do
python - $order > gpurun_out/s21/order_$(echo $order | tr -d ' ').txt 2>&1 <<'P'
import sys, torch
sys.argv = sys.argv[:1] + sys.argv[1:]
order = [int(a) for a in sys.argv[1:]]
sys.path.insert(0, '.'); sys.path.insert(0, 'omg-tools_amd')
import bench
from omgtools import workloads
dev = torch.device('cuda', 0)
problem, P = workloads.holonomic_p2p(1024)
opts = dict(tol=1e-3, max_iter=300)
for rule in order:
    r = bench.sustained_leg(problem, P, opts, 120, dev, stop_rule=bool(rule))
    print('rule', rule, 'ms/update %.4f' % r['ms_per_update'], 'solves/s %.0f' % r['solves_per_s'], flush=True)
P
done
for f in gpurun_out/s21/order_*.txt; do echo == $f; grep rule $f; done
