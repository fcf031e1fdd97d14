import sys, os; sys.path.insert(0, 'omg-tools_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import omgtools.backend as be
saved = be.create_nlp
be.create_nlp = lambda tpl, opt, name='': (None, 0.)
from omgtools.scenarios import holonomic_p2p
from omgtools.batch import BatchP2P
from oracle.nlp_numpy import NumpyNLP
problem, P = holonomic_p2p(8)
be.create_nlp = saved
tpl = problem.father.template; nlp = NumpyNLP(tpl)
opts = dict(tol=1e-6, max_iter=300)
gpu = BatchP2P(problem, P, ops='hip', options=opts, max_iter_step=300)
cpu = BatchP2P(problem, P, ops='numpy', options=opts, max_iter_step=300)
gpu.solve_cold(); cpu.solve_cold()
print('cold', gpu.host('status'), cpu.status, gpu.host('iters'), cpu.iters)
for k in range(12):
    cg = gpu.step(); cc = cpu.step()
    xg = gpu.host('x'); pg = gpu.host('p')
    fs = []
    for b in range(8):
        fg_, _ = nlp.fg(xg[b], nlp.term_coefs(pg[b])); fc_, _ = nlp.fg(cpu.x[b], nlp.term_coefs(cpu.p[b]))
        fs.append(fg_ - fc_)
    print(k, cg, 'st', gpu.host('status'), cpu.status, 'it', gpu.host('iters'), cpu.iters, 'df', np.round(fs, 6))
