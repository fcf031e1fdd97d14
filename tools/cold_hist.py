import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
import numpy as np, torch
import omgtools.backend as be
from omgtools import scenarios
from omgtools.batch import BatchP2P
name = sys.argv[1]
saved = be.create_nlp
be.create_nlp = lambda tpl, opt, name='': (None, 0.)
problem, P = getattr(scenarios, name)(1024)
be.create_nlp = saved
mpc = BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=dict(dict(P.get('solver_options', {}), **eval(os.environ.get('OMGX_OPTS', '{}'))), tol=1e-3, max_iter=300))
print('options', P.get('solver_options'))
mpc.solver.set_timing(True)
att = torch.zeros(1024, dtype=torch.int32, device='cuda')
alts = torch.stack([mpc._bent(mpc.x, s) for s in (1.0, -1.0, 2.5, -2.5)]).contiguous()
if len(sys.argv) > 2: mpc.solver.set_restarts(alts, att)
mpc._solve(False)
torch.cuda.synchronize()
print('first pass ms', mpc.solver.last_kernel_ms())
it = mpc.iters.cpu().numpy(); st = mpc.status.cpu().numpy()
print('status counts', {int(s): int((st == s).sum()) for s in np.unique(st)})
print('iters pct', np.percentile(it, [50, 90, 95, 99, 100]), 'mean', it.mean())
print('iters of failed', np.sort(it[st != 0]))
print('sum iters ok', it[st == 0].sum(), 'failed', it[st != 0].sum())
print('attempts', np.bincount(att.cpu().numpy()))
