"""Developer tool (CPU: numpy ADMM ops of the tests + oracle port as the x-update solver): the formation / rendez-vous bench
protocol of bench.py on the host -- interior-point iterations per x-update and failures, to judge a solver change on the
ADMM workloads before it goes to the GPU."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import omgtools.backend as be
from omgtools.scenarios import formation_holonomic, rendezvous_holonomic
from omgtools.admm import BatchADMM, FormationMPC
from admm_numpy_ops import NumpyAdmmOps
from oracle import port_binding

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 55
rendezvous = len(sys.argv) > 3 and sys.argv[3] == 'rendezvous'
tol = float(os.environ.get('STUDY_TOL', '1e-3'))
extra = json.loads(os.environ.get('STUDY_OPTS', '{}'))
be.create_nlp = lambda tpl, opt, name='': (None, 0.)
problem, updater, father, lay, P = (rendezvous_holonomic if rendezvous else formation_holonomic)(N)
tpl = father.template
ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'], tol=tol)
log = []
orig = port_binding.solve
def solve(*a, **k):
    k.update(extra); k['n_threads'] = 8
    r = orig(*a, **k)
    log.append((r['iters'].copy(), r['status'].copy()))
    return r
port_binding.solve = solve
admm = BatchADMM(lay, P['nbr'], ops, rho=2.0 if rendezvous else 1.0)
moving = []
for obs in problem.environment.obstacles:
    ox, ov, oa = (tpl.entry_range(obs.label, nm, 'par') for nm in ('x', 'v', 'a'))
    if np.any(P['p'][:, ov[0]:ov[1]] != 0.) or np.any(P['p'][:, oa[0]:oa[1]] != 0.):
        moving.append((ox[0], ov[0], oa[0], ox[1] - ox[0]))
mpc = FormationMPC(admm, father, tpl, lay, problem.vehicles[0], obstacles=moving, update_time=0.1, init_iter=5,
                   knot_time=problem.knot_time, consensus_is_spline=not rendezvous)
t0 = time.time()
mpc.initialize()
n_init = len(log)
cross = 0
for k in range(steps):
    status, crossed = mpc.step()
    cross += int(crossed)
it = np.array([l[0] for l in log[n_init:]]); st = np.array([l[1] for l in log[n_init:]])
print('%s %d agents tol %g opts %s: %d updates (%d crossings): mean iterations per x-update %.3f, max %d, sum over updates of the max %d, failures %d  (init: mean %.1f)  %.1f s' % (
    'rendezvous' if rendezvous else 'formation', N, tol, extra, steps, cross, it.mean(), it.max(), it.max(axis=1).sum(), (st != 0).sum(),
    np.mean([l[0].mean() for l in log[:n_init]]), time.time() - t0))
print('   statuses', dict(zip(*[a.tolist() for a in np.unique(st, return_counts=True)])), ' updates with a failure:', np.nonzero((st != 0).any(axis=1))[0][:30].tolist())
