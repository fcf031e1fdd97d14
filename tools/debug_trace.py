"""Developer tool: per-iteration trace of agent 0 of the config-2 batch (library built with -DOMGX_TRACE_DEV)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
import numpy as np
import omgtools.backend as be
from omgtools.scenarios import holonomic_p2p
be.LIB_PATH = os.path.join(ROOT, 'omg-tools_amd', 'csrc', 'libomgx_trace.so')
be.create_nlp = lambda tpl, opt, name='': (None, 0.)
problem, P = holonomic_p2p(4)
solver = be.BatchSolver(problem.father.template, 4, options=dict(tol=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-6, max_iter=80))
res = solver.solve(P['p'], P['x0'])
print(res['status'], res['iters'])
