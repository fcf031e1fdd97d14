"""Developer tool (CPU, oracle port): run the bench protocol on the host, stop at (step, agent) and print the
per-iteration trace of that one solve (port built with -DOMGX_TRACE)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd')); sys.path.insert(0, ROOT)
import numpy as np
import omgtools.backend as be
from omgtools.workloads import holonomic_p2p
from omgtools.batch import BatchP2P
from oracle import port_binding

B, step_at, agent = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
be.create_nlp = lambda tpl, opt, name='': (None, 0.)
problem, P = holonomic_p2p(B)
opts = dict(dict(tol=1e-3, max_iter=300), **json.loads(os.environ.get('STUDY_OPTS', '{}')))
mpc = BatchP2P(problem, P, ops=port_binding, options=opts)
mpc.solve_cold(bends=())
for k in range(step_at):
    mpc.step()
# the inputs of step `step_at` for `agent`: replay the glue by hand, then solve only that agent with the trace build
import copy
snap = dict(p=mpc.p.copy(), x=mpc.x.copy(), lam=mpc.lam.copy(), status=mpc.status.copy(), dw=mpc.dw.copy(), time=mpc.time)
real_solve = port_binding.solve
captured = {}
def fake(tpl, p, x, **kw):
    captured.update(p=p.copy(), x=x.copy(), kw=dict((k, v.copy() if hasattr(v, 'copy') else v) for k, v in kw.items()))
    return real_solve(tpl, p, x, **kw)
port_binding.solve = fake
crossed = mpc.step()
port_binding.solve = real_solve
print('step %d crossed %s iters of agent %d: %d (max of batch %d)' % (step_at, crossed, agent, mpc.iters[agent], mpc.iters.max()))
kw = captured['kw']
np.savez(os.path.join(ROOT, 'tools', 'scratch', 'trace_in.npz'), p=captured['p'][agent], x=captured['x'][agent], lam=kw['lam_g0'][agent],
         status=kw['status0'][agent], dw=kw['dw_state'][agent])
trace_lib = os.path.join(ROOT, 'oracle', '_build', 'libomgx_port_trace.so')
subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-pthread', '-shared', '-DOMGX_TRACE', '-Wno-unknown-pragmas', '-o', trace_lib,
                       os.path.join(ROOT, 'oracle', 'port', 'omgx_port.cpp')] + os.environ.get('TRACE_DEFS', '').split())
code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import omgtools.backend as be
from omgtools.workloads import holonomic_p2p
from oracle import port_binding
be.create_nlp = lambda tpl, opt, name='': (None, 0.)
problem, P = holonomic_p2p(1)
d = np.load(%r)
kw = %r
r = port_binding.solve(problem.father.template, d['p'][None], d['x'][None], lam_g0=d['lam'][None], status0=np.array([d['status']], dtype=np.int32),
                       dw_state=np.array([d['dw']]), **kw)
print('iters', r['iters'], 'status', r['status'])
''' % (os.path.join(ROOT, 'omg-tools_amd'), ROOT, os.path.join(ROOT, 'tools', 'scratch', 'trace_in.npz'),
       {k: v for k, v in kw.items() if k not in ('lam_g0', 'status0', 'dw_state', 'n_threads')})
env = dict(os.environ, OMGX_PORT_LIB=trace_lib)
subprocess.call([sys.executable, '-c', code], env=env)
