"""Throughput of a class with lifted auxiliaries on the device (workspace mode 6): the thirteen solves of the AGV closed loop
(tests/golden/agv_loop.npz) tiled to a batch, cold solves from the reference's warm start.   python tools/lifted_probe.py [agents]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))


def main():
    import omgtools.backend as be
    from omgtools.template import NLPTemplate
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    gold = os.path.join(ROOT, 'tests', 'golden')
    tpl = NLPTemplate.from_npz(os.path.join(gold, 'agv_fixedT.npz'))
    d = np.load(os.path.join(gold, 'agv_loop.npz'))
    idx = np.arange(n) % len(d['p'])
    p, x0 = d['p'][idx], d['x0'][idx]
    solver = be.BatchSolver(tpl, n, options=dict(tol=1e-3, max_iter=500))
    solver.set_timing(True)
    best = None
    for rep in range(3):
        t0 = time.time()
        res = solver.solve(p, x0, lbg=d['lbg'], ubg=d['ubg'])
        wall = time.time() - t0
        ms = solver.last_kernel_ms()
        best = ms if best is None else min(best, ms)
    ws = solver.workspace()
    solver.close()
    ok = (res['status'] == 0)
    same = all(np.array_equal(res['x'][k], res['x'][k % len(d['p'])]) for k in range(n))
    print(json.dumps({'class': 'AGV (fixed horizon), 381 variables / 2234 rows, 278 lifted auxiliaries', 'agents': n,
                      'solved_fraction': float(ok.mean()), 'iterations_mean': float(res['iters'].mean()), 'kernel_ms': best,
                      'solves_per_s': n / (best * 1e-3), 'iterations_per_s': float(res['iters'].sum()) / (best * 1e-3),
                      'wall_s_last': wall, 'workspace': ws, 'copies_bit_identical': bool(same)}))


if __name__ == '__main__':
    main()
