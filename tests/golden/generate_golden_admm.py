#!/usr/bin/env python
"""Golden vectors for the formation-ADMM path, produced by EXECUTING THE REFERENCE'S OWN CODE
(`problems/formation.py:33-66`, `problems/distributedproblem.py:105-169`, `problems/dualmethod.py`,
`problems/admm.py:63-168, 248-307, 313-354`) from /root/reference on the casadi stand-in of omgx_shim,
evaluated on numbers.  Scenario: `examples/formation_holonomic.py:22-44` (4 Holonomic vehicles, two
rectangles and the moving circle).  Dumped for updater 0 (tests/golden/admm_formation.npz):

  layout        order / sizes of q_i, q_ij, q_ji (which coefficients are shared, neighbour order),
  x-update NLP  f and g of `problem_upd_x` (P2P rows + augmented-Lagrangian objective, `admm.py:63-115`) at
                seeded random (x, p), with lbg / ubg and the parameter layout,
  z-update      A and b of the equality-constrained QP (`admm.py:313-354`) and the outputs z_i, z_ij of
                `problem_upd_z` (`admm.py:117-168`) for seeded random inputs, t in (0, knot_time),
  l-update      outputs of `problem_upd_l` (`admm.py:248-268`),
  residuals     outputs of `problem_upd_res` (`admm.py:270-307`).

Run in the build container only:  python tests/golden/generate_golden_admm.py"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
import omgx_shim            # noqa: E402

omgx_shim.install()
for name in ('tkinter', 'tkinter.filedialog', 'tkinter.messagebox', 'tkinter.ttk'):
    sys.modules[name] = types.ModuleType(name)


class _Any(object):
    def __init__(self, *a, **k): pass
    def __getattr__(self, n): return _Any()
    def __call__(self, *a, **k): return _Any()


tk = sys.modules['tkinter']
for n in ('Tk', 'Frame', 'Canvas', 'Button', 'Label', 'Entry', 'StringVar', 'IntVar', 'Toplevel', 'Checkbutton',
          'LabelFrame', 'Scale', 'OptionMenu', 'Radiobutton'):
    setattr(tk, n, _Any)
tk.filedialog, tk.messagebox = sys.modules['tkinter.filedialog'], sys.modules['tkinter.messagebox']
sys.path.insert(0, '/root/reference')
from omgtools import *      # noqa: E402,F401,F403   the reference
import casadi               # noqa: E402   (the stand-in)

omgx_shim.solver_factory = lambda tpl, opt: None      # nothing is solved here


def flat(v):
    return np.asarray(casadi._arr(v), float).reshape(-1, order='F')


def formation():
    N = 4
    vehicles = [Holonomic() for l in range(N)]
    fleet = Fleet(vehicles)
    configuration = RegularPolyhedron(0.2, N, np.pi / 4.).vertices.T
    init_positions = [-1.5, -1.5] + configuration
    terminal_positions = [2., 2.] + configuration
    fleet.set_configuration(configuration.tolist())
    fleet.set_initial_conditions(init_positions.tolist())
    fleet.set_terminal_conditions(terminal_positions.tolist())
    environment = Environment(room={'shape': Square(5.)})
    rectangle = Rectangle(width=3., height=0.2)
    environment.add_obstacle(Obstacle({'position': [-2.1, -0.5]}, shape=rectangle))
    environment.add_obstacle(Obstacle({'position': [1.7, -0.5]}, shape=rectangle))
    trajectories = {'velocity': {'time': [3., 4.], 'values': [[-0.15, 0.0], [0., 0.15]]}}
    environment.add_obstacle(Obstacle({'position': [1.5, 0.5]}, shape=Circle(0.4),
                                      simulation={'trajectories': trajectories}))
    problem = FormationPoint2point(fleet, environment, options={'rho': 1., 'horizon_time': 10, 'verbose': 0})
    return problem, vehicles, configuration


def rendezvous():
    """`examples/rendezvous_holonomic_export.py:31-53`: four Holonomic vehicles meet (`problems/rendezvous.py:26-67`:
    FreeEndPoint2point sub-problems, consensus on the fleet centre of the free end points)."""
    N = 4
    vehicles = [Holonomic(shapes=Circle(0.1), options={'room_constraints': None}) for l in range(N)]
    fleet = Fleet(vehicles)
    configuration = RegularPolyhedron(0.2, N, np.pi / 4.).vertices.T
    fleet.set_configuration(configuration.tolist())
    fleet.set_initial_conditions([[0., 3.], [3., 3.], [3., 0.], [0., 0.]])
    fleet.set_terminal_conditions(np.zeros((N, 2)).tolist())
    environment = Environment(room={'shape': Square(5.), 'position': [1.5, 1.5]})
    environment.add_obstacle(Obstacle({'position': [3.2, 1.0]}, shape=Rectangle(width=3., height=0.2)))
    problem = RendezVous(fleet, environment, options={'rho': 2., 'horizon_time': 10, 'init_iter': 5, 'verbose': 0})
    return problem, vehicles, configuration


def dump(problem, vehicles, configuration, out_name, seed):
    problem.init()
    upd = problem.updaters[0]
    rng = np.random.default_rng(seed)
    out = {}

    # ---- layout ---------------------------------------------------------------------------------------
    out['n_shared'] = upd.q_i_struct.shape[0]
    out['q_i_names'] = np.array(['%s/%s' % (c.label, n) for c, q in upd.q_i.items() for n in q])
    out['q_i_index'] = np.concatenate([np.asarray(i) for q in upd.q_i.values() for i in q.values()])
    out['nghb_order'] = np.array([str(n) for n in upd.q_ij.keys()])
    out['nghb_index'] = np.array([n._index for n in upd.q_ij.keys()])
    out['rel_pos_c'] = np.array([flat(v.rel_pos_c) if hasattr(v, 'rel_pos_c') else np.zeros(2) for v in vehicles])
    out['configuration'] = configuration

    # ---- x-update NLP -----------------------------------------------------------------------------------
    nlp = upd.problem_upd_x.nlp
    X, Pm = nlp['x'].cat, nlp['p'].cat
    father = upd.father_updx


    def layout_of(st):
        rows = []
        for e in st.entries:
            if e.struct is not None:
                for e2 in e.struct.entries:
                    off, shape = st.flat((e.name, e2.name))
                    rows.append((e.name + '/' + e2.name, off, shape[0], shape[1]))
            else:
                off, shape = st.flat(e.name)
                rows.append((e.name, off, shape[0], shape[1]))
        return rows


    for tag, st in (('var', nlp['x'].struct), ('par', nlp['p'].struct), ('con', nlp['g'].struct)):
        lay = layout_of(st)
        out['updx_%s_names' % tag] = np.array([r[0] for r in lay])
        out['updx_%s_layout' % tag] = np.array([r[1:] for r in lay], dtype=np.int64)
    out['updx_lb'], out['updx_ub'] = flat(father._lb.cat), flat(father._ub.cat)
    p0 = flat(father.set_parameters(0.).cat)
    x0 = flat(father.get_variables().cat)
    out['updx_x0'], out['updx_p0'] = x0, p0
    par_names = list(out['updx_par_names'])
    i_t = int(out['updx_par_layout'][par_names.index([n for n in par_names if n.endswith('/t')][0])][0])
    xs, ps, fs, gs = [], [], [], []
    for k in range(4):
        xv = x0 + 0.2 * rng.standard_normal(x0.size)
        pv = p0 + 0.1 * rng.standard_normal(p0.size) * (np.arange(p0.size) >= 0)
        # keep T, rho and the obstacle radii physical; t inside the first knot interval
        pv = np.where(np.abs(p0) > 0, p0 * (1 + 0.05 * rng.standard_normal(p0.size)), 0.3 * rng.standard_normal(p0.size))
        pv[i_t] = 0.2 * (k + 1)
        env = {X: xv.reshape(-1, 1), Pm: pv.reshape(-1, 1)}
        gv = casadi.deep(lambda: np.asarray(nlp['g'].cat.eval(env), float).reshape(-1, order='F'))
        fv = casadi.deep(lambda: float(np.asarray(casadi.MX.lift(nlp['f']).eval(env), float).reshape(-1)[0]))
        xs.append(xv); ps.append(pv); fs.append(fv); gs.append(gv)
    out.update(updx_xs=np.array(xs), updx_ps=np.array(ps), updx_fs=np.array(fs), updx_gs=np.array(gs))

    # ---- z-update: A, b and the Function ------------------------------------------------------------------
    ok, A, b = upd._check_for_lineq()
    assert ok
    par0 = upd.set_parameters_upd_z(0.)
    out['updz_par_names'] = np.array(list(upd.par_global.keys()))
    out['updz_par'] = flat(par0.cat)
    out['updz_A'] = np.asarray(A(par0.cat), float)
    out['updz_b'] = flat(b(par0.cat))
    ns, nij = upd.q_i_struct.shape[0], upd.q_ij_struct.shape[0]
    T, rho = 10., 1.3
    zin, zout = [], []
    for k in range(3):
        x_i, l_i = rng.standard_normal(ns), rng.standard_normal(ns)
        l_ij, x_j = rng.standard_normal(nij), rng.standard_normal(nij)
        t = 0.25 * (k + 1)
        o = upd.problem_upd_z(x_i, l_i, l_ij, x_j, t, T, rho, par0.cat)
        zin.append(np.r_[x_i, l_i, l_ij, x_j, t, T, rho])
        zout.append(np.r_[flat(o[0]), flat(o[1])])
    out.update(updz_in=np.array(zin), updz_out=np.array(zout))

    # ---- l-update and residuals ------------------------------------------------------------------------------
    lin, lout, rin, rout = [], [], [], []
    for k in range(3):
        x_i, z_i, l_i = (rng.standard_normal(ns) for _ in range(3))
        z_ij, l_ij, x_j = (rng.standard_normal(nij) for _ in range(3))
        o = upd.problem_upd_l(x_i, z_i, z_ij, l_i, l_ij, x_j, rho)
        lin.append(np.r_[x_i, z_i, z_ij, l_i, l_ij, x_j, rho]); lout.append(np.r_[flat(o[0]), flat(o[1])])
        z_i_p, z_ij_p = rng.standard_normal(ns), rng.standard_normal(nij)
        t = 0.3 * (k + 1)
        o = upd.problem_upd_res(x_i, z_i, z_i_p, z_ij, z_ij_p, x_j, t, T, rho)
        rin.append(np.r_[x_i, z_i, z_i_p, z_ij, z_ij_p, x_j, t, T, rho]); rout.append(np.array([float(flat(v)[0]) for v in o]))
    out.update(updl_in=np.array(lin), updl_out=np.array(lout), res_in=np.array(rin), res_out=np.array(rout))
    np.savez_compressed(os.path.join(HERE, out_name), **out)
    print('wrote ' + out_name + ': n_shared %d, neighbours %s, A %s, x-update n_var %d n_con %d n_par %d'
          % (out['n_shared'], list(out['nghb_order']), out['updz_A'].shape, x0.size, out['updx_lb'].size, p0.size))


if __name__ == '__main__':
    which = sys.argv[1:] or ['formation', 'rendezvous']
    if 'formation' in which:
        dump(*formation(), out_name='admm_formation.npz', seed=20240807 + 44)
    if 'rendezvous' in which:
        dump(*rendezvous(), out_name='admm_rendezvous.npz', seed=20240807 + 45)
