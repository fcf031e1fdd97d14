#!/usr/bin/env python
"""Per-phase cycle breakdown of ipm_solve_kernel (profiling build libomgx_prof.so,
`make -C omg-tools_amd/csrc libomgx_prof.so`).  Developer tool, not a test."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
import omgtools.backend as be

PHASES = ['jac', 'resid', 'assemble', 'factor', 'solve', 'step', 'linesearch', 'update', 'f_leaf', 'f_schur', 'f_root', 'l_A', 'l_stage', 'l_B', 'setup', 'total',
          's_desc', 's_params', 's_jac0', 's_class', 's_init', 'a_zero', 'a_pairs', 'a_rest', 'a_diag',
          'k_fwd', 'k_rootrhs', 'k_root', 'k_leafrhs', 'k_bwd', 'l_terms', 'l_rows',
          'f_park', 'f_sweep', 'f_scale', 'a_tcol', 'a_hess', 's_p_load', 's_p_div', 's_p_bspl', 's_p_slots']


def mpc_mode(problem, P, B, steps=20, warmup=3):
    """The headline protocol (bench.py): cold solve, `warmup` receding-horizon steps, then the phase counters
    of `steps` steps summed -- cycles per solve and per iteration of the warm-started steps."""
    import torch
    from omgtools.batch import BatchP2P
    mpc = BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=dict(tol=1e-3, max_iter=300))
    solver = mpc.solver
    solver.lib.omgx_batch_phase_cycles.argtypes = [C.c_void_p, C.c_void_p]
    solver.set_timing(True)
    mpc.solve_cold()
    for _ in range(warmup):
        mpc.step()
    tot = np.zeros(len(PHASES))
    its, ms = 0, []
    prof = np.zeros((B, len(PHASES)), dtype=np.int64)
    for _ in range(steps):
        mpc.step()
        torch.cuda.synchronize()
        ms.append(solver.last_kernel_ms())
        solver.lib.omgx_batch_phase_cycles(solver._h, prof.ctypes.data)
        tot += prof.sum(axis=0)
        its += int(mpc.iters.sum().item())
    n_solves = B * steps
    out = {'mode': 'mpc', 'agents': B, 'steps': steps, 'kernel_ms_mean': float(np.mean(ms)), 'kernel_ms_p50': float(np.median(ms)),
           'iters_per_solve': its / n_solves,
           'cycles_per_solve': {p: float(tot[k] / n_solves) for k, p in enumerate(PHASES)},
           'cycles_per_iter': {p: float(tot[k] / max(1, its)) for k, p in enumerate(PHASES)}}
    print(json.dumps(out, indent=1))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    be.LIB_PATH = os.environ.get('OMGX_PROF_LIB', os.path.join(ROOT, 'omg-tools_amd', 'csrc', 'libomgx_prof.so'))
    from omgtools import scenarios
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    problem, P = getattr(scenarios, os.environ.get('OMGX_SCENARIO', 'holonomic_p2p'))(B)
    be.create_nlp = saved
    tpl = problem.father.template
    if len(sys.argv) > 2 and sys.argv[2] == 'mpc':
        return mpc_mode(problem, P, B)
    solver = be.BatchSolver(tpl, B, options=dict(P.get('solver_options', {}), tol=1e-3, max_iter=300))
    warm = len(sys.argv) > 2 and sys.argv[2] == 'warm'
    for _ in range(2):
        res = solver.solve(P['p'], P['x0'])
    if warm:      # re-solve from the solution with its multipliers: the steady-state MPC case
        solver.set_options(warm_start=1)
        res = solver.solve(P['p'], res['x'], lam_g0=res['lam_g'], status0=res['status'])
    ms = solver.last_kernel_ms()
    prof = np.zeros((B, len(PHASES)), dtype=np.int64)
    solver.lib.omgx_batch_phase_cycles.argtypes = [C.c_void_p, C.c_void_p]
    solver.lib.omgx_batch_phase_cycles(solver._h, prof.ctypes.data)
    its = max(1, res['iters'].sum())
    tot = prof[:, :8].sum()
    out = {'agents': B, 'kernel_ms': ms, 'sum_iters': int(its),
           'solved': int((res['status'] == 0).sum()),
           'cycles_per_iter': {p: float(prof[:, k].sum() / its) for k, p in enumerate(PHASES)},
           'share': {p: float(prof[:, k].sum() / tot) for k, p in enumerate(PHASES)},
           'total_cycles_per_iter': float(tot / its),
           'per_agent': {'setup': float(prof[:, PHASES.index('setup')].mean()),
                         'total': float(prof[:, PHASES.index('total')].mean()),
                         'loop': float(prof[:, :8].sum(axis=1).mean()), 'iters': float(res['iters'].mean())}}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
