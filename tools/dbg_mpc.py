import sys, os; sys.path.insert(0, 'omg-tools_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import omgtools.backend as be
saved = be.create_nlp
be.create_nlp = lambda tpl, opt, name='': (None, 0.)
from omgtools.scenarios import holonomic_p2p
from omgtools.batch import BatchP2P
problem, P = holonomic_p2p(1024)
be.create_nlp = saved
mpc = BatchP2P(problem, P, ops='hip', options=dict(tol=1e-3, max_iter=300))
mpc.solve_cold()
st = mpc.host('status'); it = mpc.host('iters')
print('cold status', np.bincount(st, minlength=5), 'ms', mpc.solver.last_kernel_ms())
for k in range(12):
    cr = mpc.step()
    st = mpc.host('status'); it = mpc.host('iters')
    print(k, cr, 'status', np.bincount(st, minlength=5), 'iters ok mean %.1f max %d' % (it[st==0].mean(), it[st==0].max()), 'fail mean %.1f max %d' % (it[st!=0].mean(), it[st!=0].max()), 'sum', it.sum(), 'ms %.1f' % mpc.solver.last_kernel_ms(), 'top', np.sort(it)[-6:])
