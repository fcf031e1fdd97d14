"""Accuracy / throughput curve of the receding-horizon loop over the solver tolerance (round-5 review, item 2): for every tolerance
the closed-loop deviation from the stored SLSQP-in-the-loop reference (tools/closed_loop.py; 64 agents of config 2, 25 updates, two
knot crossings) and -- on a GPU -- the bench's headline protocol at that tolerance.

  python tools/tol_curve.py cpu             host build of the kernel source (oracle/port): the accuracy column only (the HIP path
                                            gives the same digits, tests/test_closed_loop.py)
  python tools/tol_curve.py gpu [outfile]   HIP path: accuracy + `python bench.py --tol X --no-cpu --no-extras` per tolerance
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np

TOLS = (1e-3, 3e-4, 1e-4, 3e-5, 1e-5, 1e-6)


def accuracy(make, tol):
    import closed_loop as cl
    loops = []

    def mk(problem, P, opts):
        loops.append(make(problem, P, opts))
        return IterLog(loops[-1])
    worst, first, parted, med, capped = cl.run_loop(mk, tol, 'cfg2')
    it = loops[-1]._iters
    return {'tol': tol, 'closed_loop_pos_m': float(worst[0]), 'closed_loop_vel_mps': float(worst[1]), 'closed_loop_rel': float(worst[2]),
            'median_pos_m_at_end': med, 'parted_agents': len(parted), 'capped_solves': capped,
            'mean_iters_warm': float(np.mean(it[1:])), 'mean_iters_cold': float(it[0])}


class IterLog(object):
    """Wraps a loop: mean iteration count of every solve call."""

    def __init__(self, m):
        self.m = m
        m._iters = []

    def solve_cold(self, **kw):
        r = self.m.solve_cold(**kw)
        self.m._iters.append(float(np.mean(self.m.host('iters'))))
        return r

    def step(self):
        r = self.m.step()
        self.m._iters.append(float(np.mean(self.m.host('iters'))))
        return r

    def host(self, name):
        return self.m.host(name)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'cpu'
    from omgtools.batch import BatchP2P
    rows = []
    if mode == 'cpu':
        from oracle import port_binding

        def make(problem, P, opts):
            m = BatchP2P(problem, P, ops=port_binding, options=opts)
            m.n_threads = 8
            return m
        for tol in TOLS:
            rows.append(accuracy(make, tol))
            print(json.dumps(rows[-1]))
        return
    import torch
    made = []

    def make(problem, P, opts):
        made.append(BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=opts))
        return made[-1]
    for tol in TOLS:
        row = accuracy(make, tol)
        made[-1].solver.close()
        for form, extra in (('three_streams', []), ('one_launch', ['--streams', '1'])):
            out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--tol', repr(tol), '--no-cpu', '--no-extras'] + extra,
                                 capture_output=True, text=True)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                row[form] = {'solves_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'mean_iters': d['mean_iters'],
                             'solved_fraction': d['solved_fraction'], 'max_iters_in_a_step': d['max_iters_in_a_step'],
                             'p50_batch_latency_ms': d['p50_batch_latency_ms'], 'cold_solves_per_s': d['cold_solve']['solves_per_s']}
            except Exception as e:
                row[form] = {'error': repr(e), 'stderr': out.stderr[-400:]}
        rows.append(row)
        print(json.dumps(row))
    if len(sys.argv) > 2:
        json.dump(rows, open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
