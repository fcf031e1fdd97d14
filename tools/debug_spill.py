"""Developer tool: cold solves of a small batch of the Quadrotor / 3-D classes, status and iteration counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
import numpy as np
import omgtools.backend as be
from omgtools import scenarios
be.create_nlp = lambda tpl, opt, name='': (None, 0.)
if os.environ.get('OMGX_LIB'):
    be.LIB_PATH = os.environ['OMGX_LIB']
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for scen in sys.argv[2:] or ('quadrotor_p2p', 'holonomic3d_p2p'):
    problem, P = getattr(scenarios, scen)(B)
    tpl = problem.father.template
    print(scen, {k: v for k, v in be.describe_plan(tpl).items() if k != 'order'})
    solver = be.BatchSolver(tpl, B, options=dict(P.get('solver_options', {}), tol=1e-3, max_iter=300))
    print('workspace', solver.workspace())
    for rep in range(2):
        res = solver.solve(P['p'], P['x0'])
        print(scen, 'status', np.bincount(res['status'], minlength=5), 'iters mean', res['iters'].mean(), 'max', res['iters'].max(), 'kernel ms', solver.last_kernel_ms())
    solver.close()
