"""Test helper: an `nlpsol`-shaped solver object backed by the ORACLE host port, so that the
Python host logic (Simulator / Deployer / Problem) can be exercised in the CPU test tier.  Test
infrastructure only -- the product path (`omgtools.backend.NlpSolver`) has no CPU fallback."""
import time

import numpy as np


class PortNlpSolver(object):
    def __init__(self, template, options):
        from oracle.port_binding import _backend
        options_from_problem = _backend().options_from_problem
        self.template = template
        self.options = options_from_problem(options)
        # (the same switch as omgtools.backend.NlpSolver: off by option, and off for templates that ignore `hess_approx`)
        self.fallback = bool(options.get('omgx', {}).get('hess_fallback', True)) and _backend().template_is_general(template)
        self._stats = {'return_status': 'Not_Solved', 'iter_count': 0}

    def __call__(self, x0=None, p=None, lbg=None, ubg=None, **kwargs):
        from oracle import port_binding
        from oracle.port_binding import _backend
        STATUS_STRINGS = _backend().STATUS_STRINGS
        # (the product's solver object takes a solve that gives up phase I once more with `hess_approx`: omgtools.backend.second_attempt)
        state = dict(self.options)
        fallback = self.fallback
        base = dict(state)

        def solve(p_, x_, lb_, ub_):
            return port_binding.solve(self.template, p_, x_, lb_, ub_, **state)

        def set_options(**kw):
            state.clear()
            state.update(kw)
        res = _backend().second_attempt(solve, set_options, base, fallback, np.asarray(p), np.asarray(x0), np.asarray(lbg), np.asarray(ubg))
        self._stats = {'return_status': STATUS_STRINGS[int(res['status'][0])], 'iter_count': int(res['iters'][0])}
        import os
        if os.environ.get('PORT_SOLVER_DUMP'):            # (the inputs and the result of every solve: fixtures of closed loops, tests/golden/generate_shim_fixtures.py)
            self._n = getattr(self, '_n', -1) + 1
            np.savez(os.path.join(os.environ['PORT_SOLVER_DUMP'], 'solve_%03d.npz' % self._n), p=np.asarray(p), x0=np.asarray(x0),
                     lbg=np.asarray(lbg), ubg=np.asarray(ubg), x=res['x'][0], lam_g=res['lam_g'][0], status=int(res['status'][0]),
                     iters=int(res['iters'][0]))
        return {'x': res['x'][0], 'lam_g': res['lam_g'][0]}

    def stats(self):
        return dict(self._stats)


def create_nlp(template, options, name=''):
    t0 = time.time()
    return PortNlpSolver(template, options), time.time() - t0
