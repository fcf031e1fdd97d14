"""Developer tool (CPU, oracle port as the solver object): the receding-horizon protocol of bench.py on a batch of
config-2 agents, per-step histogram of the iteration counts -- which steps have stragglers and how bad they are."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd')); sys.path.insert(0, ROOT)
import numpy as np
import omgtools.backend as be
from omgtools.workloads import holonomic_p2p
from omgtools.batch import BatchP2P
from oracle import port_binding

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 23
be.create_nlp = lambda tpl, opt, name='': (None, 0.)
problem, P = holonomic_p2p(B)
import json
mpc = BatchP2P(problem, P, ops=port_binding, options=dict(dict(tol=1e-3, max_iter=300), **json.loads(os.environ.get('STUDY_OPTS', '{}'))),
               cross_options=json.loads(os.environ.get('STUDY_CROSS', '{}')) or None)
t0 = time.time()
mpc.solve_cold(bends=())
print('cold: ok %d / %d, mean iters %.1f  (%.1f s)' % ((mpc.status == 0).sum(), B, mpc.iters.mean(), time.time() - t0))
tot = 0
tail = tail_x = 0
quiet = bool(os.environ.get('STUDY_QUIET'))
for k in range(steps):
    crossed = mpc.step()
    it = np.asarray(mpc.iters)
    tot += it.sum()
    worst = np.argsort(-it)[:4]
    tail += it.max() - 1
    tail_x += (it.max() - 1) if crossed else 0
    if not quiet or crossed: print('step %2d%s  t_rel %.2f  mean %.2f  max %2d  >2: %3d  >5: %3d  fail %d   worst agents %s' % (
        k, '*' if crossed else ' ', np.round(mpc.time, 6) % mpc.knot_time, it.mean(), it.max(), (it > 2).sum(), (it > 5).sum(),
        (np.asarray(mpc.status) != 0).sum(), list(zip(worst.tolist(), it[worst].tolist()))))
print('mean iterations per solve %.3f   tail (sum over steps of max - 1) %d, of which on crossing steps %d' % (tot / float(steps * B), tail, tail_x))

try:
    _lib = port_binding.load()
    _lib.omgx_port_cnt.restype = __import__('ctypes').c_long
    print('   counters: factorisations %d, iterations %d, line-search trials %d (%.2f per iteration), corrections %d' % (
        _lib.omgx_port_nfact(0), _lib.omgx_port_cnt(4, 0), _lib.omgx_port_cnt(8, 0), _lib.omgx_port_cnt(8, 0) / max(1.0, float(_lib.omgx_port_cnt(4, 0))), _lib.omgx_port_cnt(9, 0)))
except AttributeError:
    pass
