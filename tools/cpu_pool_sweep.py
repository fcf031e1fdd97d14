"""Scaling of the CPU baseline's pinned pool (oracle/port) with the number of threads on this host:
python tools/cpu_pool_sweep.py [agents]  -- prints solves/s of the bench protocol per pool size and the
cgroup cpu quota, so that bench.py's `cpu_baseline.cores` can be read against what the box really grants."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    import omgtools.backend as be
    from omgtools.scenarios import holonomic_p2p
    from omgtools.batch import BatchP2P
    from oracle import port_binding
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
        if os.path.exists(path):
            print(path, open(path).read().strip())
    cpus = port_binding.physical_cpus()
    print('logical cpus', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'physical', len(cpus))
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    problem, P = holonomic_p2p(n)
    opts = dict(tol=1e-3, max_iter=300)
    sizes = sorted(set([1, 2, 4, 8, 16, 32, 64, len(cpus)]))
    for nt in [s for s in sizes if s <= len(cpus)]:
        pool = port_binding.PortPool(problem.father.template, cpus=cpus[:nt])
        sub = {'p': P['p'], 'x0': P['x0']} if nt > 1 else {'p': P['p'][:64], 'x0': P['x0'][:64]}
        mpc = BatchP2P(problem, sub, ops=port_binding, options=opts)
        mpc.pool = pool
        t0 = time.perf_counter()
        mpc.solve_cold()
        t_cold = time.perf_counter() - t0
        for _ in range(3):
            mpc.step()
        ok, t0 = 0, time.perf_counter()
        for _ in range(10):
            mpc.step()
            ok += int((mpc.status == 0).sum())
        dt = time.perf_counter() - t0
        print('threads %4d: %9.0f solves/s warm (%.2f ms/step), cold %8.0f solves/s' % (nt, ok / dt, dt / 10 * 1e3, mpc.B / t_cold))
        pool.close()


if __name__ == '__main__':
    main()
