"""Solver parity against an independent NLP solver (CasADi/IPOPT outputs are
unobtainable, so scipy SLSQP on the restated problem is the reference): from the reference's
initial guess both reach the same local minimum -- objective to 1.5e-5 relative, trajectory
coefficients (the output the reference consumes) to 1e-4 (solver tolerance 3e-6: at 1e-6 the last barrier
problem, mu = 1e-7, is solved inside the rounding noise and the outcome depends on the order of the sums).  CPU tier: the host port; GPU tier:
the HIP path through the C ABI (tests/test_gpu_solver.py reuses `slsqp_cases`)."""
import numpy as np
import pytest


def slsqp_cases():
    """[(name, template, spline slice, p, x0, x_slsqp, f_slsqp)] for config 1 (the reference's
    `examples/p2p_holonomic.py`) and the first agents of the synthetic config-2 batch."""
    import omgtools.backend as be
    from oracle.nlp_numpy import NumpyNLP
    from slsqp_reference import solve_slsqp
    from test_golden_nlp import build
    from omgtools.scenarios import holonomic_p2p
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        out = []
        pr = build('cfg1_p2p_holonomic')
        pr.reinitialize()
        fa, tpl = pr.father, pr.father.template
        x0 = np.asarray(fa.get_variables()).reshape(-1)
        p = fa.set_parameters(0.).cat.copy()
        sl = slice(*tpl.entry_range(pr.vehicles[0].label, 'splines_seg0', 'var'))
        out.append(('cfg1', tpl, sl, p, x0) + solve_slsqp(NumpyNLP(tpl), tpl, x0, p))
        pr2, P = holonomic_p2p(3)
        tpl2 = pr2.father.template
        sl2 = slice(*tpl2.entry_range(pr2.vehicles[0].label, 'splines_seg0', 'var'))
        nlp2 = NumpyNLP(tpl2)
        for b in range(3):
            out.append(('cfg2[%d]' % b, tpl2, sl2, P['p'][b], P['x0'][b]) + solve_slsqp(nlp2, tpl2, P['x0'][b], P['p'][b]))
        return out
    finally:
        be.create_nlp = saved


def check_against_slsqp(cases, solve):
    from oracle.nlp_numpy import NumpyNLP
    matched = 0
    for name, tpl, sl, p, x0, xs, fs, ok in cases:
        res = solve(tpl, p, x0)
        if res['status'][0] != 0:
            # (at tight tolerances a few per cent of the solves end in the rounding
            # noise of the last iterations; which ones depends on the order of floating-point sums
            assert res['status'][0] == 4, name
            continue
        nlp = NumpyNLP(tpl)
        f = nlp.fg(res['x'][0], nlp.term_coefs(p))[0]
        if not ok or abs(fs - f) > 1e-4 * (1 + abs(f)):
            continue                    # SLSQP failed or went to another local minimum: not comparable
        assert abs(fs - f) < 1.5e-5 * (1 + abs(f)), name
        assert np.abs(res['x'][0][sl] - xs[sl]).max() < 1e-4, name
        matched += 1
    assert matched >= 3


def test_port_reaches_the_slsqp_minimum():
    from oracle import port_binding
    check_against_slsqp(slsqp_cases(),
                        lambda tpl, p, x0: port_binding.solve(tpl, p[None], x0[None], tol=3e-6, max_iter=500))
