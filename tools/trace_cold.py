"""Developer tool (CPU, oracle port built with -DOMGX_TRACE): per-iteration trace of the cold solve of one config-2 agent."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd')); sys.path.insert(0, ROOT)
trace_lib = os.path.join(ROOT, 'oracle', '_build', 'libomgx_port_trace.so')
if 'OMGX_PORT_LIB' not in os.environ:
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-fPIC', '-pthread', '-shared', '-DOMGX_TRACE', '-Wno-unknown-pragmas', '-o', trace_lib,
                           os.path.join(ROOT, 'oracle', 'port', 'omgx_port.cpp')] + os.environ.get('TRACE_DEFS', '').split())
    os.execve(sys.executable, [sys.executable] + sys.argv, dict(os.environ, OMGX_PORT_LIB=trace_lib))
import numpy as np
import omgtools.backend as be
from omgtools.scenarios import holonomic_p2p
from oracle import port_binding
B, agent = int(sys.argv[1]), int(sys.argv[2])
be.create_nlp = lambda tpl, opt, name='': (None, 0.)
problem, P = holonomic_p2p(B)
opts = dict(dict(tol=1e-3, max_iter=300), **json.loads(os.environ.get('STUDY_OPTS', '{}')))
r = port_binding.solve(problem.father.template, P['p'][agent:agent + 1], P['x0'][agent:agent + 1], **opts)
print('iters', r['iters'], 'status', r['status'])
