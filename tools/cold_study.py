"""Developer tool (CPU, oracle port): cold solves of the config-2 batch, iteration-count percentiles."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd')); sys.path.insert(0, ROOT)
import numpy as np
import omgtools.backend as be
import omgtools.scenarios as scen
from oracle import port_binding
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
be.create_nlp = lambda tpl, opt, name='': (None, 0.)
problem, P = getattr(scen, sys.argv[2] if len(sys.argv) > 2 else 'holonomic_p2p')(B)
opts = dict(dict(tol=1e-3, max_iter=300), **json.loads(os.environ.get('STUDY_OPTS', '{}')))
t0 = time.time()
r = port_binding.solve(problem.father.template, P['p'], P['x0'], n_threads=8, **opts)
it = r['iters']
print('cold %d agents: ok %d, mean %.2f p50 %d p90 %d p99 %d max %d  worst %s  (%.1f s)' % (
    B, (r['status'] == 0).sum(), it.mean(), np.percentile(it, 50), np.percentile(it, 90), np.percentile(it, 99), it.max(),
    list(zip(np.argsort(-it)[:6].tolist(), np.sort(it)[::-1][:6].tolist())), time.time() - t0))

try:
    _lib = port_binding.load()
    _lib.omgx_port_cnt.restype = __import__('ctypes').c_long
    print('   counters: factorisations %d, iterations %d, line-search trials %d (%.2f per iteration), corrections %d' % (
        _lib.omgx_port_nfact(0), _lib.omgx_port_cnt(4, 0), _lib.omgx_port_cnt(8, 0), _lib.omgx_port_cnt(8, 0) / max(1.0, float(_lib.omgx_port_cnt(4, 0))), _lib.omgx_port_cnt(9, 0)))
except AttributeError:
    pass
