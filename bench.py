#!/usr/bin/env python
"""bench.py -- MPC solves/sec of the batched spline-MPC solve path on MI355X.

Workload (BASELINE.json configs[1]): 1024-agent Holonomic Point2point batch per
GPU, degree-3 B-spline, knot_intervals=11, 3 circular obstacles, fp64; seeded
synthetic scenarios (omgtools/scenarios.py).  Protocol of SURVEY.md
§8d: a cold solve of every agent from the reference's initial guess, then
receding-horizon MPC steps (update_time 0.1 s, ideal prediction, warm start).  One
timed "step" = one MPC step of the whole batch: prediction + horizon bookkeeping +
`omgx_batch_solve`, everything resident in HBM.  Solver tolerance = the
reference's default `ipopt.tol = 1e-3` (`problems/problem.py:57`).  The cold-solve
rate is reported alongside (`cold_solve`).

Launch: python bench.py --gpus N --steps K --warmup W   (N>1 via torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
sys.path.insert(0, ROOT)

# One process per GPU under a process group (the driver's launch for N > 1): RCCL brings streams of its own, and the HIP runtime maps
# streams onto four hardware queues per priority level.  The sub-batch streams of the per-step product path have a level to themselves
# (omgtools.batch.sub_batch_streams: 2.20 M solves/s with four or eight queues under a one-rank group, profiles/r06_stream_placement.txt);
# eight queues are kept for such launches as headroom for whatever a multi-rank communicator creates (unmeasured here: one GPU per box).
# Must be in the environment before the runtime starts, i.e. before torch is imported.
if int(os.environ.get('WORLD_SIZE', '1')) > 1 or os.environ.get('OMGX_FORCE_DIST') == '1':
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch

FP64_MATRIX_PEAK_TFLOPS = 78.6     # MI355X datasheet FP64 matrix (MFMA f64) peak
HBM_PEAK_GBPS = 8000.0             # MI355X HBM3E (MI355X_MICROARCH.md)


def emit(line):
    """The ONE JSON line, as the LAST line of stdout: whatever native libraries left in the C stdio buffer (RCCL prints a version
    banner to stdout when its first communicator comes up) goes out before it."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(line) + '\n')
    sys.stdout.flush()


def quiet_host():
    """Before a timed region: collect now and keep the cyclic collector out of the region.  A generation-2 collection of this
    process (torch + numpy loaded: 35 - 60 ms) in the middle of a loop whose host side runs ahead of the device drains the
    queue: measured on the rendez-vous bench as 1.2 vs 1.7 - 2.4 ms per step from run to run, the pause landing on a different
    step with every change of the allocation history (first process on a box: no .pyc yet, another history, no pause)."""
    gc.collect()
    gc.disable()


def executed_flops_per_iter(tpl):
    """Flops one interior-point iteration of the structured solver actually executes on the linear algebra (SURVEY.md 8d
    asks for them next to the dense-n figure): per hyperplane leaf the banded LDL' (n B (B + 1)), the forward substitution of
    its carried rows (coupling rows + right-hand side, n (2 B + 1) each), its Schur complement W D^-1 W' (symmetric, 2 n per
    entry); the root block nr^3 / 3; the substitutions; three flops per Jacobian pair of J' Sigma J.  One factorisation
    attempt per iteration (the inertia correction adds about 0.3 on cold solves).  From `omgx_plan_describe`."""
    from omgtools.backend import describe_plan
    pl = describe_plan(tpl)
    nr = pl['n_root'] + pl['n_eq']
    fl = nr ** 3 / 3.0 + 2.0 * nr ** 2
    for n, bw, nc in zip(pl['leaf_sizes'], pl['leaf_bw'], pl['leaf_cpl']):
        B = min(bw, n - 1)
        if B == 0:                                      # diagonal leaf: a division and a few products per coupling entry
            fl += n * (4 + 2 * 3)
            continue
        rows = nc + 1
        fl += n * B * (B + 1) + rows * n * (2 * B + 1) + (rows * (rows + 1) // 2) * 2 * n + rows * n
        fl += 2 * nc * n + 2 * n * B                    # leaf substitutions
    fl += 3.0 * pl['n_pairs']
    return fl


def sample_roofline(torch, tpl, veh, dev, n_agents, horizon_time, reps=20):
    """Second roofline of SURVEY 8d: post-solve trajectory sampling (A11, `sample_kernel`) against HBM
    bandwidth.  Algorithmic bytes per agent = 8 n_spl L (coefficients in) + 8 n_der n_spl n_samp (samples
    out), n_samp = 1001, fp64.  Timed with events on the launch stream; outside the headline's timed
    region."""
    from omgtools.backend import BatchSolver
    sol = BatchSolver(tpl, n_agents, device=dev.index or 0, options=dict(tol=1e-3, max_iter=1))
    sol.set_stream(torch.cuda.current_stream().cuda_stream)
    n_spl, L, n_der, n_samp = veh.n_spl, len(veh.basis), 3, 1001
    lo = tpl.entry_range(veh.label, 'splines_seg0', 'var')[0]
    x = torch.randn((n_agents, tpl.n_var), dtype=torch.float64, device=dev)
    t0 = torch.zeros(n_agents, dtype=torch.float64, device=dev)
    out = torch.empty((n_agents, n_der, n_spl, n_samp), dtype=torch.float64, device=dev)
    knots = veh.basis.knots * horizon_time
    dt = horizon_time / (n_samp - 1)
    for _ in range(3):
        sol.sample(x, lo, n_spl, veh.degree, knots, n_der, t0, dt, n_samp, out=out, device=True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    # the kernel's own begin / end stamps (events carried by its dispatch packet): a pair recorded around the call also
    # times ~35 us of launch path, more than the 1024-agent kernel itself
    for a, b in ev:
        a.record(); b.record()
        sol.set_launch_events(a, b)
        sol.sample(x, lo, n_spl, veh.degree, knots, n_der, t0, dt, n_samp, out=out, device=True)
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    around = []
    for a, b in ev:
        a.record()
        sol.sample(x, lo, n_spl, veh.degree, knots, n_der, t0, dt, n_samp, out=out, device=True)
        b.record()
    torch.cuda.synchronize()
    call_ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    nbytes = n_agents * 8.0 * (n_spl * L + n_der * n_spl * n_samp)
    # calibration: a plain fill of the same output buffer (what a pure write stream reaches on this box)
    for a, b in ev:
        a.record()
        out.fill_(1.0)
        b.record()
    torch.cuda.synchronize()
    fill_ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    sol.close()
    return {'bound': 'hbm', 'kernel': 'sample_kernel', 'agents': n_agents, 'achieved': nbytes / (ms * 1e-3) / 1e9,
            'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            'bytes_per_launch': nbytes, 'kernel_ms': ms, 'events_around_the_call_ms': call_ms,
            'fill_same_buffer_GBps': out.numel() * 8.0 / (fill_ms * 1e-3) / 1e9}


def measured_traffic(n_agents):
    """(HBM bytes per launch of ipm_solve_kernel over receding-horizon steps, source file) from the COMMITTED PMC passes
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this bench at 1024 agents; the first launches are the cold
    solves, the rest warm steps) -- NOT measured in this run: `roofline.traffic_source` names the file.  (None, None) for
    other batch sizes.  Counter unit KB; FETCH_SIZE doubled: on gfx950 rocprofv3 reports half the bytes of wide coalesced reads
    (MI355X_MICROARCH.md, HBM section) -- the table records are 16-byte-per-lane loads -- so this is an upper bound for the
    mixed access widths of this kernel; WRITE_SIZE as reported."""
    for name in ('r06_pmc_hbm.json', 'r05_pmc_hbm.json', 'r04_pmc_hbm.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if n_agents == 1024 and os.path.exists(path):
            d = json.load(open(path))
            n_cold = int(d.get('cold_launches', 4))
            warm = lambda key: float(np.mean(d[key]['per_launch_kb'][n_cold:])) * 1024.0 * float(d.get('launches_per_step', 1))
            return 2.0 * warm('FETCH_SIZE') + warm('WRITE_SIZE'), 'profiles/' + name
    return None, None


def cpu_baseline(problem, P, opts, steps, warmup, budget_s):
    """Oracle CPU port on the same workload, protocol and tolerance: cold solve, `warmup` untimed
    receding-horizon steps, then `steps` timed ones (exactly the GPU's timed region), repeated from
    the cold start until about `budget_s` seconds of timed CPU work are collected.  The host solver has
    the life cycle of the HIP handle (oracle/port_binding.PortPool): structure plan built once, worker
    threads created once and pinned to one logical cpu per physical core, the glue of a step (prediction,
    obstacle motion, knot shift) executed per agent by the worker that then solves it -- no Python between
    the agents of a step.  Second leg: one pinned thread on a 64-agent sample."""
    from omgtools.batch import BatchP2P
    from oracle import port_binding            # the checker, timed as the reported CPU baseline
    cpus = port_binding.granted_cpus()
    quota = port_binding.cpu_quota()
    tpl = problem.father.template
    out = {}
    for label, use, n_agents, budget in (('all', cpus, P['p'].shape[0], budget_s), ('one', cpus[:1], min(64, P['p'].shape[0]), budget_s / 4.)):
        sub = {'p': P['p'][:n_agents], 'x0': P['x0'][:n_agents]}
        pool = port_binding.PortPool(tpl, cpus=use)
        ok, its, dt, reps = 0, 0, 0.0, 0
        while dt < budget and reps < 200:
            mpc = BatchP2P(problem, sub, ops=port_binding, options=opts)
            mpc.pool = pool
            mpc.solve_cold()
            for _ in range(warmup):
                mpc.step()
            t0 = time.perf_counter()
            for _ in range(steps):
                mpc.step()
                ok += int((mpc.status == 0).sum())
                its += int(mpc.iters.sum())
            dt += time.perf_counter() - t0
            reps += 1
        pool.close()
        out[label] = dict(rate=ok / dt, reps=reps, agents=n_agents, dt=dt, iters=its / float(reps * steps * n_agents))
    a, o = out['all'], out['one']
    try:
        slsqp = slsqp_leg(problem, P, opts, budget_s / 2.)
    except Exception as e:                        # (a side figure must never cost the line)
        slsqp = {'error': repr(e)}
    return {'slsqp_restatement': slsqp, 'value': a['rate'], 'unit': 'solves/s', 'cores': len(cpus), 'kind': 'port',
            'sample': '%d repetitions of the bench protocol (cold solve, %d warm-up steps untimed; %d receding-horizon '
                      'steps timed) on all %d agents with a persistent pool of %d threads pinned one per physical core '
                      '(%d logical cpus visible, cgroup cpu quota %s: more threads than the quota only get throttled, '
                      'profiles/r02_cpu_pool_sweep.txt), step glue in C inside the workers, %.1f s timed; single pinned '
                      'thread: %d repetitions on the first %d agents, %.1f s timed'
                      % (a['reps'], warmup, steps, a['agents'], len(cpus), os.cpu_count() or 1,
                         'none' if quota is None else '%.0f cpus' % quota, a['dt'], o['reps'], o['agents'], o['dt']),
            'single_thread_value': o['rate'], 'mean_iters': a['iters'],
            'note': 'host build of the same interior-point statements (oracle/port), not IPOPT: CasADi/IPOPT is not '
                    'installable here (BASELINE.md)'}


def slsqp_leg(problem, P, opts, budget_s):
    """BASELINE.md section 3 / SURVEY 8d's second CPU figure: the numpy restatement of the NLP (oracle/nlp_numpy.py) solved by
    scipy SLSQP -- an independent dense SQP, single thread -- on the first agents of the same workload under the same
    protocol (cold solve from the reference's guess, then warm-started receding-horizon steps), for about `budget_s` seconds.
    Not IPOPT, not tuned: context for the port figure, never a target."""
    from omgtools.batch import BatchP2P
    from oracle.nlp_numpy import NumpyNLP
    from oracle.slsqp_numpy import solve_slsqp

    class Slsqp(object):                          # (the `ops` shape BatchP2P's host protocol expects)
        def __init__(self, tpl):
            self.nlp = NumpyNLP(tpl)
            self.calls, self.ok, self.t = 0, 0, 0.0

        def solve(self, tpl, p, x, **kw):
            B = p.shape[0]
            xs, st = np.array(x, dtype=float), np.zeros(B, dtype=np.int32)
            for b in range(B):
                t0 = time.perf_counter()
                xb, _, ok = solve_slsqp(self.nlp, tpl, x[b], p[b], maxiter=200, accept=(0, 8), viol_tol=1e-6)
                self.t += time.perf_counter() - t0
                xs[b] = xb
                st[b] = 0 if ok else 1
                self.calls += 1
                self.ok += int(ok)
            return {'x': xs, 'lam_g': np.zeros((B, tpl.n_con)), 'status': st, 'iters': np.zeros(B, dtype=np.int32)}

    tpl = problem.father.template
    n, cold, warm = 0, Slsqp(tpl), Slsqp(tpl)
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < budget_s and n < P['p'].shape[0]:
        sub = {'p': P['p'][n:n + 1], 'x0': P['x0'][n:n + 1]}
        mpc = BatchP2P(problem, sub, ops=cold, options=opts)
        mpc.solve_cold()
        mpc.port = warm
        for _ in range(3):
            mpc.step()
        n += 1
    return {'cold_solves_per_s': cold.ok / max(cold.t, 1e-9), 'warm_solves_per_s': warm.ok / max(warm.t, 1e-9), 'cores': 1,
            'agents': n, 'cold_solves': cold.calls, 'warm_solves': warm.calls, 'solved': cold.ok + warm.ok,
            'seconds': cold.t + warm.t,
            'note': 'numpy restatement of the NLP + scipy SLSQP (ftol 1e-12: it has no tolerance that corresponds to ipopt.tol), one '
                    'thread, the first agents of the same workload: cold solve + 3 warm-started receding-horizon steps each'}


def latency_episodes(mpc, x0_init, p_init, n_ep, n_steps, host):
    """p50 over n_ep x n_steps timed receding-horizon steps (median over >= 200 timed batch solves);
    every episode restarts from the cold solve so that the agents are under way, not parked at the goal.
    host=False: the span is the solve kernel (events on its stream), data resident.  host=True: the span is what
    a caller with host buffers pays (8d's "device time incl. parameter upload and coefficient download"): p and
    the warm start uploaded from pinned host memory, the step, then x, the multipliers' status and iteration
    counts downloaded to pinned host memory."""
    B = mpc.B
    ms, ok = [], 0
    if host:
        p_h = torch.empty(mpc.p.shape, dtype=torch.float64).pin_memory()
        x_h = torch.empty(mpc.x.shape, dtype=torch.float64).pin_memory()
        st_h = torch.empty(B, dtype=torch.int32).pin_memory()
        it_h = torch.empty(B, dtype=torch.int32).pin_memory()
    evs = []
    for e in range(n_ep):
        mpc.x.copy_(x0_init)
        mpc.p.copy_(p_init)
        mpc.time = 0.0
        mpc.solve_cold()
        for k in range(n_steps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if host:
                p_h.copy_(mpc.p)
                x_h.copy_(mpc.x)
                torch.cuda.synchronize()
                a.record()
                mpc.p.copy_(p_h, non_blocking=True)
                mpc.x.copy_(x_h, non_blocking=True)
                mpc.step()
                x_h.copy_(mpc.x, non_blocking=True)
                st_h.copy_(mpc.status, non_blocking=True)
                it_h.copy_(mpc.iters, non_blocking=True)
                b.record()
                torch.cuda.synchronize()
                ok += int((st_h == 0).sum())
            else:
                a.record()                  # (creates the handles; the library re-stamps both on the solve kernel's dispatch)
                b.record()
                mpc.step(events=(a, b))
                ok += int((mpc.status == 0).sum().item())
            evs.append((a, b))
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in evs]
    return {'samples': len(ms), 'p50_ms': float(np.median(ms)), 'p90_ms': float(np.percentile(ms, 90)),
            'max_ms': float(np.max(ms)), 'mean_ms': float(np.mean(ms)), 'solves_per_s': ok / (sum(ms) * 1e-3),
            'solved_fraction': ok / float(len(ms) * B)}


def store_leg(mpc, problem, tpl, x0_init, p_init, n_steps, dev):
    """The same receding-horizon steps with `Vehicle.store` fused behind the solve (`omgx_batch_set_store`, A11:
    state, input, dinput and v_tot of every agent on its 1001-point sample grid, written by the solve kernel straight
    from the solution): wall time per step between barriers, against the same steps without it."""
    veh = problem.vehicles[0]
    T = float(problem.options['horizon_time'])
    B, nd, n_samp, sample_time = mpc.B, veh.n_dim, 1001, 0.01
    lo = tpl.entry_range(veh.label, 'splines_seg0', 'var')[0]
    f64 = dict(dtype=torch.float64, device=dev)
    out, vt, t0 = torch.zeros((B, 3, nd, n_samp), **f64), torch.zeros((B, n_samp), **f64), torch.zeros(B, **f64)
    res = {}
    for name in ('plain', 'with_store'):
        mpc.x.copy_(x0_init); mpc.p.copy_(p_init); mpc.time = 0.0
        mpc.solve_cold()
        if name == 'with_store':
            mpc.solver.set_store(out, vt, t0, lo, nd, veh.degree, veh.basis.knots, 3, n_samp, sample_time / T, 1.0 / T)
        for _ in range(3):
            mpc.step()
        torch.cuda.synchronize()
        t_0 = time.perf_counter()
        for _ in range(n_steps):
            mpc.step()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t_0) / n_steps * 1e3
        ok = int((mpc.status == 0).sum().item())
        mpc.solver.set_store(None)
    bytes_out = out.numel() * 8 + vt.numel() * 8
    return {'ms_per_step_plain': res['plain'], 'ms_per_step_with_store': res['with_store'],
            'store_MB_per_step': bytes_out / 1e6, 'solves_per_s_with_store': ok / (res['with_store'] * 1e-3),
            'note': 'state / input / dinput / v_tot of every agent on 1001 samples, written by the solve kernel (A11 fused)'}


def lifted_leg(n_agents):
    """SURVEY 8(f)3 on the device: a class whose rows are products of more than four variable factors (the reference's AGV model,
    `vehicles/agv.py:50`, written with lifted auxiliaries -- DESIGN.md 2 -- 381 variables / 2234 rows, workspace mode 6): the thirteen
    solves of its closed loop tiled to a batch, cold solves from the reference's warm start.  One launch, timed by its own events."""
    from omgtools.workloads import agv_loop
    from omgtools.backend import BatchSolver
    tpl, P = agv_loop(n_agents)
    solver = BatchSolver(tpl, n_agents, options=dict(tol=1e-3, max_iter=500))
    try:
        solver.set_timing(True)
        res = solver.solve(P['p'], P['x0'], lbg=P['lbg'], ubg=P['ubg'])
        ms = solver.last_kernel_ms()
        ws = solver.workspace()
    finally:
        solver.close()
    its = np.asarray(res['iters'])
    return {'class': 'AGV, fixed horizon (vehicles/agv.py:50): 381 variables (278 lifted auxiliaries), 2234 rows', 'agents': n_agents,
            'solved_fraction': float((np.asarray(res['status']) == 0).mean()), 'mean_iters': float(its.mean()),
            'same_iterations_as_host_build': bool(np.array_equal(its, P['iters_host'])), 'kernel_ms': ms,
            'solves_per_s': n_agents / (ms * 1e-3), 'iterations_per_s': float(its.sum()) / (ms * 1e-3), 'workspace_mode': ws['mode'],
            'lds_bytes_per_agent': ws['lds_bytes'],
            'note': 'cold solves of the thirteen problems of the AGV closed loop, tiled; functional, not fast: the auxiliaries sit in a dense root of order 579 kept in the slab'}


def rollout_leg(mpc, x0_init, p_init, n_steps, warmup, dev):
    """The same protocol -- cold solve, `warmup` steps, `n_steps` timed steps -- with the timed steps in ONE launch
    (`omgx_batch_rollout`): every agent runs its own predict / shift / solve loop without the barrier between the steps of
    different agents (they are independent problems).  Bit-identical plans to the stepwise loop (tests/test_gpu_rollout.py).
    Not the headline: a deployment that feeds measured states back needs the per-step launch."""
    B = mpc.B
    mpc.x.copy_(x0_init); mpc.p.copy_(p_init); mpc.time = 0.0
    mpc.solve_cold(bends=())
    for _ in range(warmup):
        mpc.step()
    stats = torch.zeros((n_steps, 4), dtype=torch.int64, device=dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); b.record()
    torch.cuda.synchronize()
    mpc.solver.set_stats(stats)
    mpc.solver.set_launch_events(a, b)
    quiet_host()
    t_0 = time.perf_counter()
    crossings = mpc.rollout(n_steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_0
    gc.enable()
    mpc.solver.set_stats(None)
    st = stats.cpu().numpy()
    assert (st[:, 3] == B).all()
    n_ok = float(st[:, 0].sum()) / n_steps
    return {'solves_per_s': n_ok * n_steps / wall, 'ms_per_step': wall / n_steps * 1e3, 'kernel_ms_per_step': a.elapsed_time(b) / n_steps,
            'steps': n_steps, 'warmup': warmup, 'solved_fraction': n_ok / B, 'mean_iters': float(st[:, 1].sum()) / (n_steps * B),
            'step_max_iters': [int(v) for v in st[:, 2]], 'knot_crossings': crossings,
            'note': 'the timed steps in one launch (omgx_batch_rollout): no barrier between the steps of different agents; the same bits as the stepwise loop'}


def stepwise_leg(problem, P, opts, n_steps, warmup, dev, n_streams):
    """The headline protocol on the OTHER form of the per-step path (one handle when the headline runs sub-batch launches, and
    two sub-batches otherwise): wall time between host syncs, solved agents counted per step by the solve kernels
    (`omgx_batch_set_stats` on every handle)."""
    from omgtools.batch import StreamedP2P, receding_horizon_batch
    B = P['p'].shape[0]
    rh = receding_horizon_batch(problem, P, device=dev, n_streams=n_streams, options=opts)
    parts = rh.parts if isinstance(rh, StreamedP2P) else [rh]
    rh.solve_cold(bends=())
    for _ in range(warmup):
        rh.step()
    torch.cuda.synchronize()
    stats = [torch.zeros((n_steps, 4), dtype=torch.int64, device=dev) for _ in parts]
    for m, sd in zip(parts, stats):
        m.solver.set_stats(sd)
    quiet_host()
    t_0 = time.perf_counter()
    for _ in range(n_steps):
        rh.step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_0
    gc.enable()
    st = sum(sd.cpu().numpy() for sd in stats)
    st[:, 2] = np.max([sd.cpu().numpy()[:, 2] for sd in stats], axis=0)
    for m in parts:
        m.solver.set_stats(None)
        m.solver.close()
    assert (st[:, 3] == B).all()
    return {'solves_per_s': float(st[:, 0].sum()) / wall, 'max_iters_in_a_step': int(st[:, 2].max()), 'ms_per_step': wall / n_steps * 1e3, 'steps': n_steps, 'warmup': warmup,
            'solved_fraction': float(st[:, 0].sum()) / (n_steps * B), 'mean_iters': float(st[:, 1].sum()) / (n_steps * B),
            'launches_per_step': len(parts),
            'note': '%d sub-batch(es) of %d agents, per-step launches' % (len(parts), B // len(parts))}


def host_boundary_pipelined(problem, P, opts, n_steps, warmup, dev, engine='kernel', depth=4):
    """SURVEY 8d's span as worded -- "device time incl. parameter upload and coefficient download" -- as a THROUGHPUT: the
    caller keeps p and x in (pinned) host memory.  Per step and per half of the batch, on that half's own HIP stream:
    the step's glue on the device (prediction: stands for the plant), the parameters DOWN to the host (the caller's copy of
    the measured state), the parameters UP again from pinned memory, the warm-started solve, then x, status and iteration
    counts DOWN into one of `depth` pinned buffer sets.  Nothing syncs the host inside a step: it waits for the event of step
    k - depth before it hands buffer set k % depth out again (a ring: the host consumes step k - depth + 1 .. k - 1 while step k
    runs; depth 2 left the device queue two steps deep and every wake-up of the blocked host thread showed), and the
    transfers of one half overlap the solve of the other.  Solved agents are counted from the downloaded status words.
    engine='kernel': the transfers are `omgx_batch_transfer` launches (the kernel reads / writes the pinned buffers over the
    host link: no hand-over to a copy engine); engine='memcpy': hipMemcpyAsync through torch (`copy_(non_blocking=True)`)."""
    from omgtools.batch import StreamedP2P
    B = P['p'].shape[0]
    n_streams = 2 if B % 2 == 0 else 1
    rh = StreamedP2P(problem, P, n_streams=n_streams, device=dev, options=opts)
    n = B // n_streams
    tpl = rh.tpl
    pin = lambda *shape, **kw: torch.empty(shape, **kw).pin_memory()
    p_h = [pin(n, tpl.n_par, dtype=torch.float64) for _ in rh.parts]
    bufs = [[dict(x=pin(n, tpl.n_var, dtype=torch.float64), status=pin(n, dtype=torch.int32), iters=pin(n, dtype=torch.int32))
             for _ in range(depth)] for _ in rh.parts]
    done = [[None] * depth for _ in rh.parts]
    rh.solve_cold(bends=())
    for _ in range(warmup):
        rh.step()
    rh.synchronize()
    kern = engine in ('kernel', 'mapped')
    if engine == 'mapped':
        # the parameters LIVE in pinned host memory (mapped into the device's address space): the prediction kernel writes them
        # there over the host link -- the caller's copy of the measured state -- and the solve kernel reads them from there:
        # upload and download of p without a transfer of their own (static obstacles: no torch arithmetic touches p)
        for kp, m in enumerate(rh.parts):
            p_h[kp].copy_(m.p)
            m.p = p_h[kp]
        torch.cuda.synchronize()

    plans = {}

    def plan(m, key, pairs):                               # (prepared transfers: the pointer checks once, one library call per use)
        if key not in plans:
            plans[key] = m.solver.transfer_plan(pairs)
        return plans[key]

    def hook(k_part):
        def up_down(m):
            if engine == 'mapped':
                return
            if kern:
                plan(m, (k_part, 'down'), [(p_h[k_part], m.p)]).run()      # the caller's copy of the parameters (measured state) ...
                plan(m, (k_part, 'up'), [(m.p, p_h[k_part])]).run()        # ... and their upload: what the solve reads came over the host link
            else:
                p_h[k_part].copy_(m.p, non_blocking=True)
                m.p.copy_(p_h[k_part], non_blocking=True)
        return up_down
    hooks = [hook(k) for k in range(len(rh.parts))]
    ok = 0
    quiet_host()
    t_0 = time.perf_counter()
    for k in range(n_steps):
        s = k % depth
        for kp in range(len(rh.parts)):                    # buffer set s was handed out at step k - depth: its consumer is done when ...
            if done[kp][s] is not None:
                done[kp][s].synchronize()                  # ... that step's downloads have landed
                ok += int((bufs[kp][s]['status'] == 0).sum())
        rh.step(before_solve=hooks)
        for kp, (m, st) in enumerate(zip(rh.parts, rh.streams)):
            with torch.cuda.stream(st):
                bf = bufs[kp][s]
                if kern:                                     # (m.x alternates between two buffers: a plan per buffer)
                    plan(m, (kp, s, m.x.data_ptr()), [(bf['x'], m.x), (bf['status'], m.status), (bf['iters'], m.iters)]).run()
                else:
                    bf['x'].copy_(m.x, non_blocking=True)
                    bf['status'].copy_(m.status, non_blocking=True)
                    bf['iters'].copy_(m.iters, non_blocking=True)
                done[kp][s] = st.record_event()
    t_host = time.perf_counter() - t_0
    rh.synchronize()
    wall = time.perf_counter() - t_0
    gc.enable()
    for k in range(max(0, n_steps - depth), n_steps):     # the last steps' buffers
        for kp in range(len(rh.parts)):
            ok += int((bufs[kp][k % depth]['status'] == 0).sum())
    rh.close()
    mb = (2 * tpl.n_par + tpl.n_var + 1) * 8 * B / 1e6
    return {'solves_per_s': ok / wall, 'ms_per_step': wall / n_steps * 1e3, 'steps': n_steps, 'warmup': warmup,
            'solved_fraction': ok / float(n_steps * B), 'host_link_MB_per_step': mb, 'engine': engine, 'ring_depth': depth,
            'host_enqueue_ms_per_step': t_host / n_steps * 1e3,
            'note': 'per step and half-batch on its own stream: p down + p up (pinned), solve, x / status / iters down into double-buffered '
                    'pinned memory; no host sync inside a step (the host waits for step k - depth before reusing a buffer set of the ring); transfers by '
                    + ({'kernel': 'omgx_batch_transfer kernels (no copy engine)', 'memcpy': 'hipMemcpyAsync (copy engine)',
                        'mapped': 'p lives in pinned host memory (written by the prediction kernel, read by the solve kernel over the host '
                                  'link: no transfer of its own), x / status / iters by one omgx_batch_transfer kernel'}[engine])}


def closed_loop_parity(opts, dev):
    """`parity_at_tol`: what the solver settings of this line mean for the plans -- the closed loop of the first 64 agents of the
    workload (cold solve, 25 updates with two knot crossings, every update predicted from the product's OWN previous plan) against the
    stored loop of an independent solver (scipy SLSQP in the loop, tests/golden/closed_loop_cfg2.npz; generator
    tests/golden/generate_closed_loop.py), in the form of the reference's replay test (`export/tests/point2point/test.cpp:116-141`:
    sampled state and input after every update), two-sided.  A fixture (data) and numpy, no oracle code; the same comparison as
    tests/test_closed_loop.py (tools/closed_loop.py)."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import closed_loop as cl
    from omgtools.batch import BatchP2P
    made = []

    def make(problem, P, o):
        made.append(BatchP2P(problem, P, ops='hip', device=dev, options=dict(o, **opts)))
        return made[-1]
    try:
        worst, first, parted, med, capped = cl.run_loop(make, opts['tol'], 'cfg2')
    finally:
        for m in made:
            m.solver.close()
    return {'tol': opts['tol'], 'compl_inf_tol': opts.get('compl_inf_tol', 0.0), 'constr_viol_tol': opts.get('constr_viol_tol', 0.0),
            'closed_loop_pos_m': float(worst[0]), 'closed_loop_vel_mps': float(worst[1]), 'closed_loop_rel': float(worst[2]),
            'median_pos_m_at_end': med, 'after_cold_solve_pos_m': float(first[0]), 'parted_agents': len(parted), 'solves_at_iteration_cap': capped,
            'agents': 64, 'updates': 25,
            'reference': 'scipy SLSQP in the loop (tests/golden/closed_loop_cfg2.npz), sampled position / velocity of the first 0.2 s of '
                         'every new plan, two-sided; rel = |a - b| / max(|a|, |b|, 0.1)'}


def tolerance_leg(problem, P, opts, n_steps, warmup, dev):
    """The headline protocol (per-step product path) at other solver settings, with the closed-loop figures of the same settings."""
    from omgtools.batch import BatchP2P
    out = stepwise_leg(problem, P, opts, n_steps, warmup, dev, 'auto')
    out.pop('note', None)
    # the same steps as ONE launch (omgx_batch_rollout): no barrier between the steps of different agents -- a straggler of a knot
    # crossing then delays only itself, where the per-step path makes every agent of its sub-batch wait for it
    mpc = BatchP2P(problem, P, ops='hip', device=dev, options=opts)
    try:
        ro = rollout_leg(mpc, mpc.x.clone(), mpc.p.clone(), n_steps, warmup, dev)
        out['rollout'] = {k: ro[k] for k in ('solves_per_s', 'ms_per_step', 'solved_fraction', 'mean_iters')}
    except Exception as e:
        out['rollout'] = {'error': repr(e)}
    finally:
        mpc.solver.close()
    out.update(parity=closed_loop_parity(opts, dev))
    return out


def sustained_leg(problem, P, opts, n_steps, dev, stop_tol=1e-3, stop_rule=True):
    """The whole manoeuvre (round-5 review, item 4): the reference's loop runs until `stop_criterium` (`execution/simulator.py:39-62`,
    `problems/point2point.py:98-102` -> `vehicles/holonomic.py:145-151`: |state - poseT| <= stop_tol and |input| <= stop_tol, stop_tol
    = 1e-3, `vehicles/vehicle.py:72`) -- ~100 updates at T = 10 s -- where the headline times updates 4 - 23.  Here: cold solve, then
    `n_steps` updates of the batch on the per-step product path without a host sync.  stop_rule (default): every vehicle's loop ends
    where the reference's does -- the solve kernel tests the criterion and does not solve an agent again once it held
    (`omgx_batch_set_stop`); the solves are counted by the kernels (launch statistics).  Without it every agent is solved at every
    update and an agent counts only while the logged states say it is under way (the figure of the first version of this leg)."""
    from omgtools.batch import StreamedP2P, receding_horizon_batch
    B = P['p'].shape[0]
    rh = receding_horizon_batch(problem, P, device=dev, n_streams='auto', options=opts)
    parts = rh.parts if isinstance(rh, StreamedP2P) else [rh]
    veh, tpl = problem.vehicles[0], problem.father.template
    nd = veh.n_dim
    o_pose = tpl.entry_range(veh.label, 'poseT', 'par')[0]
    rh.solve_cold(bends=())
    if stop_rule:
        rh.stop_at_arrival(stop_tol)
    torch.cuda.synchronize()
    stats = [torch.zeros((n_steps, 4), dtype=torch.int64, device=dev) for _ in parts]
    logs = [torch.zeros((n_steps, m.B, 2 * nd), dtype=torch.float64, device=dev) for m in parts]
    count = [0] * len(parts)

    def hook(kp):
        def log_state(m):                                   # (on the sub-batch's own stream, between the prediction and the solve)
            k = count[kp]
            logs[kp][k, :, :nd].copy_(m.p[:, m.o_state0:m.o_state0 + nd])
            logs[kp][k, :, nd:].copy_(m.p[:, m.o_input0:m.o_input0 + nd])
            count[kp] = k + 1
        return log_state
    hooks = [hook(k) for k in range(len(parts))]
    for m, sd in zip(parts, stats):
        m.solver.set_stats(sd)
    quiet_host()
    t_0 = time.perf_counter()
    for _ in range(n_steps):
        rh.step(before_solve=hooks if len(parts) > 1 else hooks[0])
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_0
    gc.enable()
    st = sum(sd.cpu().numpy() for sd in stats).astype(float)
    st[:, 2] = np.max([sd.cpu().numpy()[:, 2] for sd in stats], axis=0)
    log = torch.cat(logs, dim=1).cpu().numpy()              # [n_steps, B, 2 nd]
    pose = torch.cat([m.p[:, o_pose:o_pose + nd] for m in parts]).cpu().numpy()
    for m in parts:
        m.solver.set_stats(None)
        m.solver.close()
    there = (np.linalg.norm(log[:, :, :nd] - pose[None], axis=2) <= stop_tol) & (np.linalg.norm(log[:, :, nd:], axis=2) <= stop_tol)
    arrived_at = np.where(there.any(axis=0), there.argmax(axis=0), n_steps)        # first update at which the criterion holds
    under_way = (np.arange(n_steps)[:, None] < arrived_at[None, :]).sum(axis=1)    # agents the reference's loop would still be solving
    solved_by_kernels = st[:, 3]                                                    # agents the launches of an update solved
    counted = solved_by_kernels if stop_rule else under_way
    win = 20
    windows = [{'updates': '%d-%d' % (k, min(k + win, n_steps) - 1), 'mean_iters': float(st[k:k + win, 1].sum() / max(1.0, st[k:k + win, 3].sum())),
                'max_iters': int(st[k:k + win, 2].max()), 'agents_under_way': int(counted[k])} for k in range(0, n_steps, win)]
    at_arrival = log[np.minimum(arrived_at, n_steps - 1), np.arange(B), :nd]
    dist = np.linalg.norm(at_arrival - pose, axis=1)
    last_busy = int(np.flatnonzero(counted > 0).max()) + 1 if (counted > 0).any() else 0
    return {'solves_per_s': float(counted.sum()) / wall, 'solves': int(counted.sum()), 'stop_rule': bool(stop_rule),
            'solves_by_the_logged_states': int(under_way.sum()),
            'ms_per_update': wall / n_steps * 1e3, 'updates': n_steps, 'updates_with_agents_under_way': last_busy,
            'solved_fraction': float(st[:, 0].sum()) / max(1.0, st[:, 3].sum()),
            'mean_iters': float(st[:, 1].sum()) / max(1.0, st[:, 3].sum()), 'max_iters': int(st[:, 2].max()),
            'arrived_fraction': float((arrived_at < n_steps).mean()), 'updates_to_arrival_p50': float(np.median(arrived_at)),
            'updates_to_arrival_max': int(arrived_at.max()), 'distance_to_goal_at_arrival_max_m': float(dist.max()),
            'launches_per_update': len(parts), 'stop_tol': stop_tol, 'windows': windows,
            'note': ('cold solve, then every update of the manoeuvre on the per-step product path; every vehicle\'s loop ends at the update its '
                     'state meets the reference\'s stop criterion (tested by the solve kernel, omgx_batch_set_stop); solves counted by the kernels')
                    if stop_rule else
                    ('cold solve, then every update of the manoeuvre on the per-step product path; every agent is solved at every update, '
                     'solves_per_s counts an agent only until the logged states meet the reference\'s stop criterion')}


def manoeuvre_rollout(problem, P, opts, n_steps, dev, stop_tol=1e-3):
    """The whole manoeuvre as ONE launch (`omgx_batch_rollout` with the stop rule on): every agent runs its own loop -- prediction,
    knot-crossing shift, warm-started solve -- until its state meets the reference's stop criterion, at most `n_steps` updates, without
    the barrier between the updates of different agents.  For simulation / evaluation runs with ideal prediction (what `Simulator.run`
    does for one vehicle); the same bits per agent as the per-step loop (tests/test_gpu_rollout.py)."""
    from omgtools.batch import BatchP2P
    m = BatchP2P(problem, P, ops='hip', device=dev, options=opts)
    try:
        m.solve_cold(bends=())
        m.stop_at_arrival(stop_tol)
        stats = torch.zeros((n_steps, 4), dtype=torch.int64, device=dev)
        m.solver.set_stats(stats)
        torch.cuda.synchronize()
        quiet_host()
        t_0 = time.perf_counter()
        m.rollout(n_steps)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t_0
        gc.enable()
        m.solver.set_stats(None)
        st = stats.cpu().numpy().astype(float)
        return {'solves_per_s': float(st[:, 3].sum()) / wall, 'solves': int(st[:, 3].sum()), 'ms': wall * 1e3, 'solved_fraction': float(st[:, 0].sum()) / max(1.0, st[:, 3].sum()),
                'mean_iters': float(st[:, 1].sum()) / max(1.0, st[:, 3].sum()), 'max_iters': int(st[:, 2].max()),
                'agents_under_way_at_the_end': int(m.under_way.sum().item()),
                'note': 'one launch for the whole manoeuvre: every agent loops until its state meets the stop criterion'}
    finally:
        m.solver.close()


def without_solver_objects(fn, *a, **kw):
    """A front-end builder called for its template only (`Point2point.init` would create a solver object of its own)."""
    import omgtools.backend as be
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        return fn(*a, **kw)
    finally:
        be.create_nlp = saved


def p2p_workload(args, seed, gap=0.25):
    """BASELINE configs[1] from its committed problem bundle (`omgtools.workloads`: template + seeded parameters, no front-end
    module imported); other knot counts / obstacle numbers (developer options) are built through the front end."""
    if args.knot_intervals == 11 and args.obstacles == 3:
        from omgtools import workloads
        return workloads.holonomic_p2p(args.agents, seed=seed, gap=gap)
    from omgtools import scenarios
    return without_solver_objects(scenarios.holonomic_p2p, args.agents, knot_intervals=args.knot_intervals, n_obs=args.obstacles,
                                  seed=seed, gap=gap)


def unedited_rule(args, dev, seed):
    """SURVEY 8d's obstacle rule as written (discs rejected only when they overlap; omgtools/scenarios.py adds
    a passable-gap rule for the headline workload): solved fractions of the same protocol on that generator."""
    from omgtools.batch import BatchP2P
    problem, P = p2p_workload(args, seed, gap=0.0)
    mpc = BatchP2P(problem, P, ops='hip', device=dev, options=dict(tol=args.tol, max_iter=300))
    mpc.solve_cold()
    cold_ok = int((mpc.status == 0).sum().item())
    ok = 0
    for _ in range(20):
        mpc.step()
        ok += int((mpc.status == 0).sum().item())
    mpc.solver.close()
    return {'rule': 'discs may touch (no passable-gap rejection)', 'cold_solved_fraction': cold_ok / float(args.agents),
            'step_solved_fraction': ok / float(20 * args.agents), 'steps': 20}


def bench_formation(args, rank, local_rank, world, dist, dev):
    """configs[3]: FormationPoint2point ADMM, 512 Holonomic agents (fixed total: strong scaling).  The reference's
    receding-horizon protocol (`problems/dualmethod.py:200-224`, `problems/admm.py:477-491, 584-628`): `init_iter` = 5
    iterations at the start time, then one step = one update: time advanced by update_time = 0.1 s, the initial
    conditions predicted from the current plan (on the device), the moving obstacle advanced, the knot-crossing shift of
    x and of the consensus state, and ONE ADMM iteration (x-update + exchanges + z / l / residuals)."""
    from omgtools import workloads
    from omgtools.backend import BatchSolver
    from omgtools.admm import BatchADMM, HipAdmmOps, FormationMPC
    from omgtools.distributed import shard_range, reduce_report
    N = 512 if args.agents == 1024 else args.agents
    rendezvous = args.workload == 'rendezvous'
    # (the problem bundle of the fleet size, omgtools/data; another size: the front end builds its template)
    if workloads.have(('rendezvous' if rendezvous else 'formation') + '_holonomic_k10_%d' % N):
        problem, updater, father, lay, P = (workloads.rendezvous_holonomic if rendezvous else workloads.formation_holonomic)(N)
    else:
        from omgtools import scenarios
        problem, updater, father, lay, P = without_solver_objects(scenarios.rendezvous_holonomic if rendezvous else scenarios.formation_holonomic, N)
    tpl = father.template
    lo, hi = shard_range(N, rank, world)
    solver = BatchSolver(tpl, hi - lo, device=local_rank, options=dict(tol=args.tol, max_iter=300))
    ops = HipAdmmOps(solver, tpl, lay, P['p'][lo:hi], P['x0'][lo:hi], dev)
    admm = BatchADMM(lay, P['nbr'], ops, rank=rank, world=world, dist=dist if world > 1 else None, rho=2.0 if rendezvous else 1.0)
    moving = []
    for obs in problem.environment.obstacles:
        ox, ov, oa = (tpl.entry_range(obs.label, nm, 'par') for nm in ('x', 'v', 'a'))
        if np.any(P['p'][:, ov[0]:ov[1]] != 0.) or np.any(P['p'][:, oa[0]:oa[1]] != 0.):
            moving.append((ox[0], ov[0], oa[0], ox[1] - ox[0]))
    mpc = FormationMPC(admm, father, tpl, lay, problem.vehicles[0], obstacles=moving, update_time=0.1, init_iter=5,
                       knot_time=problem.knot_time, consensus_is_spline=not rendezvous)
    mpc.initialize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    crossings = 0
    for _ in range(args.warmup):
        mpc.step()
    barrier()
    stats = torch.zeros((args.steps, 4), dtype=torch.int64, device=dev)     # per x-update: solved, sum / max of iterations, agents
    solver.set_stats(stats)
    quiet_host()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        status, crossed = mpc.step()                        # nothing leaves the device inside the loop
        crossings += int(crossed)
    t_host = time.perf_counter() - t0                      # (enqueue time of the timed steps: far below `elapsed` unless the host is the bound)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    solver.set_stats(None)
    # where an update's time goes (a second, untimed pass with an event after every phase of the iteration: x-update,
    # centre, the two collectives, z / lambda update, read-back -- what a multi-GPU run needs to be diagnosable)
    ops.timeline = []
    for _ in range(min(args.steps, 20)):
        mpc.step()
    barrier()
    phases = ops.phase_times()
    ops.timeline = None
    stats = stats.cpu().numpy()
    if os.environ.get('OMGX_PHASES'):                      # developer: per-phase cycles of the last x-update (profiling build)
        import ctypes
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools'))
        from phase_profile import PHASES as _PH
        prof = np.zeros((hi - lo, len(_PH)), dtype=np.int64)
        solver.lib.omgx_batch_phase_cycles.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        solver.lib.omgx_batch_phase_cycles(solver._h, prof.ctypes.data)
        print({p_: round(float(prof[:, k].mean()) / 1e3, 1) for k, p_ in enumerate(_PH) if prof[:, k].mean() > 500}, file=sys.stderr)
    n_ok = int((status == 0).sum().item())
    res = admm.residuals[-1]
    elapsed, n_ok_all = reduce_report(elapsed, n_ok, device=dev, dist=dist)
    if rank != 0:
        return
    emit({
        'metric': 'ADMM agent-updates/sec, %d-agent Holonomic %s' % (N, 'RendezVous' if rendezvous else 'FormationPoint2point'),
        'value': n_ok_all * args.steps / elapsed,
        'unit': 'agent-updates/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': ('RendezVous ADMM (free end points, `problems/rendezvous.py`), %d Holonomic agents, circular '
                                'interconnection, knot_intervals=10, 1 rectangular obstacle, rho=2, tol=%g' if rendezvous else
                                'configs[3]: FormationPoint2point ADMM, %d Holonomic agents, circular '
                                'interconnection, knot_intervals=10, 2 rectangular obstacles + 1 moving circle, rho=1, tol=%g')
                               % (N, args.tol), 'agents_total': N, 'n_var': tpl.n_var, 'n_con': tpl.n_con,
                   'parallelism': 'agents sharded contiguously; two all_gathers per iteration (x_i rows; [z_ij | l_ij] rows + residual sums)'},
        'solved_fraction': n_ok_all / float(N), 'residuals': list(res), 'knot_crossings_in_timed_steps': crossings,
        'protocol': 'init_iter=5, then per step: update_time 0.1 s, device-side prediction, moving obstacle advanced, knot-crossing shift, 1 ADMM iteration',
        'x_update_mean_iters': float(stats[:, 1].sum()) / max(1, int(stats[:, 3].sum())), 'x_update_max_iters': int(stats[:, 2].max()),
        'phase_ms': dict((k, round(v, 4)) for k, v in phases.items()), 'host_enqueue_ms_per_step': t_host / args.steps * 1e3,
        'phase_note': 'rank 0, mean over a separate untimed pass of the same protocol with an event after every phase of the iteration; x_update includes the prediction / shift glue of the step'})


def bench_cold(args, rank, local_rank, world, dist, dev):
    """configs[2] / configs[4] (parity-test configurations, not the headline): cold solves of a
    batch of Quadrotor (K=13, 5 moving circles) or Holonomic3D (K=15, 10 spheres) agents.  Their
    per-agent arrays exceed one CU's LDS; the library spills to HBM slabs (omgx_batch_workspace)."""
    from omgtools import workloads
    from omgtools.backend import BatchSolver
    from omgtools.distributed import reduce_report
    fn = {'quadrotor': workloads.quadrotor_p2p, 'holonomic3d': workloads.holonomic3d_p2p}[args.workload]
    B = args.agents
    problem, P = fn(B, seed=20240807 + (3 if args.workload == 'quadrotor' else 5) + 1000 * rank)
    tpl = problem.father.template
    from omgtools.batch import BatchP2P
    mpc = BatchP2P(problem, P, ops='hip', device=dev, options=dict(P.get('solver_options', {}), tol=args.tol, max_iter=300))
    x0_init, p_init = mpc.x.clone(), mpc.p.clone()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    # one step = the cold solve of the whole batch from the reference's initial guess, restarts of the agents whose
    # phase I stalls included (BatchP2P.solve_cold: restart guesses handed to the launch)
    passes = 0
    for k in range(args.warmup + args.steps):
        if k == args.warmup:
            barrier()
            quiet_host()
            t0 = time.perf_counter()
        mpc.x.copy_(x0_init)
        mpc.p.copy_(p_init)
        mpc.time = 0.0
        passes = mpc.solve_cold()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [elapsed / args.steps * 1e3]
    status, iters = mpc.status, mpc.iters
    n_ok = int((status == 0).sum().item())
    it_sum = int(iters.sum().item())
    elapsed, n_ok_all = reduce_report(elapsed, n_ok, device=dev, dist=dist)
    # receding-horizon steps of the same batch (the protocol of the headline: cold solve, warm-up, timed steps)
    rh_steps = max(args.steps, 10)
    for _ in range(args.warmup):
        mpc.step()
    barrier()
    quiet_host()
    t0 = time.perf_counter()
    rh_ok, rh_it = 0, 0
    st_log = torch.zeros((rh_steps, B), dtype=torch.int32, device=dev)
    it_log = torch.zeros((rh_steps, B), dtype=torch.int32, device=dev)
    for k in range(rh_steps):
        mpc.step()
        st_log[k].copy_(mpc.status)
        it_log[k].copy_(mpc.iters)
    barrier()
    rh_elapsed = time.perf_counter() - t0
    rh_ok, rh_it = int((st_log == 0).sum().item()), int(it_log.sum().item())
    rh_elapsed, rh_ok_all = reduce_report(rh_elapsed, rh_ok, device=dev, dist=dist)
    if rank != 0:
        return
    n = tpl.n_var
    flops = it_sum * (n ** 3 / 3.0 + 2.0 * n ** 2)
    k_ms = float(np.mean(kernel_ms))
    emit({
        'metric': 'cold MPC solves/sec, %s batch' % args.workload, 'value': n_ok_all * args.steps / elapsed,
        'unit': 'solves/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%s: %d agents per GPU, cold solve from the reference initial guess (agents that do not '
                               'converge from it are solved again, inside the same launch, from the guess bent sideways: '
                               'at most %d restarts per agent), tol=%g'
                               % (args.workload, B, passes, args.tol), 'n_var': tpl.n_var, 'n_con': tpl.n_con},
        'solved_fraction': n_ok / float(B), 'mean_iters': it_sum / float(B), 'workspace': mpc.solver.workspace(),
        'receding_horizon': {'solves_per_s': rh_ok_all / rh_elapsed, 'ms_per_step': rh_elapsed / rh_steps * 1e3,
                             'steps': rh_steps, 'warmup': args.warmup, 'solved_fraction': rh_ok / float(rh_steps * B),
                             'mean_iters': rh_it / float(rh_steps * B), 'max_iters': int(it_log.max().item()),
                             'protocol': 'cold solve, warm-up steps, then timed steps (update_time 0.1 s, ideal '
                                         'prediction incl. the second derivative for the Quadrotor, moving obstacles '
                                         'advanced, primal-dual warm start)'},
        'roofline': {'bound': 'mfma', 'kernel': 'ipm_solve_kernel', 'achieved': flops / (k_ms * 1e-3) / 1e12,
                     'peak': FP64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': flops / (k_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS, 'traffic': None,
                     'kernel_ms': k_ms, 'note': 'iterations of the final pass of every agent; time = the whole cold step',
                     'executed_flops_per_iter': executed_flops_per_iter(tpl), 'dense_n_flops_per_iter': tpl.n_var ** 3 / 3.0 + 2.0 * tpl.n_var ** 2,
                     'executed_TFLOPs': it_sum * executed_flops_per_iter(tpl) / (k_ms * 1e-3) / 1e12,
                     'frac_executed': it_sum * executed_flops_per_iter(tpl) / (k_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                     'read_this': 'executed_TFLOPs / frac_executed are what the structured factorisation does; `achieved` / `frac` follow SURVEY 8d\'s '
                                  'dense-n convention (n^3 / 3 + 2 n^2 per iteration), which at this n_var flatters by %.0fx'
                                  % ((tpl.n_var ** 3 / 3.0 + 2.0 * tpl.n_var ** 2) / executed_flops_per_iter(tpl))}})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--agents', type=int, default=1024, help='agents per GPU')
    ap.add_argument('--tol', type=float, default=1e-3)
    ap.add_argument('--knot-intervals', type=int, default=11, help='(p2p workload; BASELINE configs[1]: 11)')
    ap.add_argument('--obstacles', type=int, default=3, help='(p2p workload; BASELINE configs[1]: 3)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0,
                    help='seconds of timed CPU work for the cpu_baseline leg (all host cores; a quarter of it on one)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the latency / host-boundary / 8d-rule legs (profiling runs)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help="p2p workload at N > 1: weak = --agents per GPU (default), strong = --agents in total, sharded over the ranks "
                         "(BASELINE's metric as worded: the 1024-agent batch at 1/2/4/8 GPU)")
    ap.add_argument('--streams', type=int, default=0,
                    help='p2p: sub-batches of the per-step path on separate HIP streams (0 = automatic: three -- '
                         'omgtools.batch.PRODUCT_PATH_STREAMS -- when the batch is at least two rounds of resident workgroups, else one)')
    ap.add_argument('--ipopt-defaults', action='store_true',
                    help="also test IPOPT's absolute tolerances on the unscaled problem at their documented defaults (compl_inf_tol = "
                         "constr_viol_tol = 1e-4, in force when the reference sets only ipopt.tol): omgx_options version 8")
    ap.add_argument('--refine', type=int, default=None, help='p2p: omgx_options.refine (default: the library default, 1); 0 switches the refinement of regularised steps off (A/B runs)')
    ap.add_argument('--no-parity', action='store_true', help='skip the closed-loop parity leg (`parity_at_tol`: 26 launches of a 64-agent batch) -- profiling passes whose per-launch averages must cover the 1024-agent launches only')
    ap.add_argument('--sustained-steps', type=int, default=120, help='updates of the whole-manoeuvre leg (`sustained`)')
    ap.add_argument('--workload', choices=['p2p', 'formation', 'rendezvous', 'quadrotor', 'holonomic3d'], default='p2p',
                    help="p2p = BASELINE.json configs[1] (headline); formation = configs[3], 512-agent ADMM")
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    dist = None
    # (OMGX_FORCE_DIST=1: the process group, the barriers and the report's all_reduces also with one rank -- the RCCL
    # path of `--gpus N` exercised on a one-GPU box, e.g. under `python -m torch.distributed.run --nproc-per-node 1`)
    if world > 1 or os.environ.get('OMGX_FORCE_DIST') == '1':
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29533'),
                              RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    if args.workload in ('formation', 'rendezvous'):
        return bench_formation(args, rank, local_rank, world, dist, dev)
    if args.workload in ('quadrotor', 'holonomic3d'):
        return bench_cold(args, rank, local_rank, world, dist, dev)
    from omgtools.batch import BatchP2P, StreamedP2P, receding_horizon_batch
    from omgtools.distributed import reduce_report, shard_workload
    strong = args.scaling == 'strong'
    if strong:
        # BASELINE's metric as worded: "1024-agent Holonomic P2P at 1/2/4/8 GPU" -- the SAME 1024 agents, sharded contiguously
        # over the ranks (128 per GPU at N = 8: below one round of resident workgroups, the regime of the kernel's latency floor)
        problem, P_all = p2p_workload(args, 20240807 + 2)
        P, (lo, hi) = shard_workload(P_all, rank, world)
    else:
        problem, P = p2p_workload(args, 20240807 + 2 + 1000 * rank)
    B = P['p'].shape[0]
    tpl = problem.father.template
    opts = dict(tol=args.tol, max_iter=300)
    if args.ipopt_defaults:
        from omgtools.backend import IPOPT_DEFAULT_TOLERANCES
        opts.update(IPOPT_DEFAULT_TOLERANCES)
    if args.refine is not None:
        opts['refine'] = args.refine
    # single-handle batch: the cold co-headline and the side legs (latency, fused store, rollout)
    mpc = BatchP2P(problem, P, ops='hip', device=dev, options=opts)
    solver = mpc.solver

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- cold solve from the reference's initial guess (reported, not the headline) ----------
    x0_init, p_init = mpc.x.clone(), mpc.p.clone()
    cold_ms, cold_first_ms, cold_passes, cold_kernel_all = [], [], 0, []
    solver.set_timing(True)
    for rep in range(4):                    # (the first repetition is not counted: lazy loading of the small torch kernels)
        mpc.x.copy_(x0_init)
        mpc.p.copy_(p_init)
        mpc.time = 0.0
        # the whole cold solve between two events on the launch stream: the first pass over every agent and the
        # restart passes for the (rare) agents whose phase I stalls from the reference's straight-line guess
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        mpc.solve_cold(bends=())
        cold_first_ms.append(solver.last_kernel_ms())
        cold_kernel_all.append(cold_first_ms[-1])
        first_ok = int((mpc.status == 0).sum().item())
        cold_passes = mpc.restart_failed()
        b.record()
        torch.cuda.synchronize()
        cold_ms.append(a.elapsed_time(b))
        if rep == 0:
            cold_ms, cold_first_ms = [], []
    cold_ok = int((mpc.status == 0).sum().item())
    cold_iters = int(mpc.iters.sum().item())
    solver.set_timing(False)
    # ---- receding-horizon steps: SURVEY.md 8d protocol (cold solve, then warm-started steps) ----
    # THE PER-STEP PRODUCT PATH (`receding_horizon_batch`): one handle, or -- for a batch of at least two rounds of resident
    # workgroups, the headline's 1024 agents on 512 slots -- three stream-ordered sub-batch launches per step (round 5; per agent the
    # same launches and bits as the single handle, tests/test_gpu_rollout.py).  `--streams 1` forces the single handle.
    rh = receding_horizon_batch(problem, P, device=dev, n_streams='auto' if args.streams == 0 else args.streams, options=opts)
    parts = rh.parts if isinstance(rh, StreamedP2P) else [rh]
    n_parts = len(parts)
    for m in parts:
        m.solver.set_timing(True)
    rh.solve_cold(bends=())
    torch.cuda.synchronize()
    rh_cold_ms = [m.solver.last_kernel_ms() for m in parts]
    rh_cold_iters = sum(int(m.iters.sum().item()) for m in parts)
    # touch the knot-crossing path once (lazy kernel loading, allocator) outside the timed region, then restore the cold solution
    for m, st in zip(parts, getattr(rh, 'streams', [torch.cuda.current_stream()])):
        m.solver.set_timing(False)          # the timed steps carry the bench's own events on the solve kernel's dispatch
        with torch.cuda.stream(st):
            x_sol, lam_sol = m.x.clone(), m.lam.clone()
            m._shift()
            m.x.copy_(x_sol)
            m.lam = lam_sol
    torch.cuda.synchronize()
    # per-step statistics are logged on the device (no host sync inside the timed region): every solve kernel carries two
    # events on its own dispatch packet (begin / end stamps of the kernel itself, nothing extra on the stream), the per-step
    # counts (solved agents, sum and maximum of the iteration counts) are added up by the solve kernel itself in a
    # [W + K, 4] device array per handle (omgx_batch_set_stats)
    K, W = args.steps, args.warmup
    stats_d = [torch.zeros((W + K, 4), dtype=torch.int64, device=dev) for _ in parts]
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in parts] for _ in range(W + K)]
    ev_base = torch.cuda.Event(enable_timing=True)
    for row in ev:                          # (torch creates the HIP event on the first record: their handles go to the library)
        for a, b in row:
            a.record()
            b.record()
    torch.cuda.synchronize()
    for m, sd in zip(parts, stats_d):
        m.solver.set_stats(sd)
    for k in range(W + K):
        if k == W:
            barrier()
            quiet_host()
            ev_base.record()
            t0 = time.perf_counter()
        if n_parts > 1:
            rh.step(events=ev[k])
        else:
            rh.step(events=ev[k][0])
    t_host = time.perf_counter() - t0                      # (enqueue time of the timed steps: below `elapsed` unless the host is the bound)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    for m in parts:
        m.solver.set_stats(None)
    stats = sum(sd.cpu().numpy() for sd in stats_d)   # rows: {solved, sum of iterations, agents}; the maximum separately
    stats[:, 2] = np.max([sd.cpu().numpy()[:, 2] for sd in stats_d], axis=0)
    assert (stats[:, 3] == B).all()
    # launch intervals on the device clock (ms since ev_base): what the solve kernels occupied the chip for is the UNION of
    # the intervals (the sub-batch launches of one step, and the next step of a sub-batch, overlap)
    spans = [[(ev_base.elapsed_time(a), ev_base.elapsed_time(b)) for a, b in row] for row in ev[W:]]

    def union_ms(iv):
        tot, end = 0.0, -1e300
        for lo_, hi_ in sorted(iv):
            if hi_ > end:
                tot += hi_ - max(lo_, end)
                end = hi_
        return tot
    busy_ms = union_ms([iv for row in spans for iv in row])
    step_latency_ms = [max(h for _, h in row) - min(l for l, _ in row) for row in spans]     # first kernel start -> last kernel end of the step
    all_ms = [a.elapsed_time(b) for row in ev for a, b in row]
    n_ok = float(stats[W:, 0].sum()) / K                       # solved agents per step (mean)
    it_sum = int(stats[W:, 1].sum())
    n_meas = K
    # every launch of the solve kernel in this process (what `rocprofv3 --stats` averages over)
    launches_ms = cold_kernel_all + rh_cold_ms + all_ms            # (restart passes, if any, are further launches: not in this list)
    launches_iters = len(cold_kernel_all) * cold_iters + rh_cold_iters + int(stats[:, 1].sum())
    elapsed, n_ok_all = reduce_report(elapsed, n_ok, device=dev, dist=dist)
    if rank != 0:
        return
    value = n_ok_all * args.steps / elapsed
    k_ms = busy_ms / K                                     # chip time of the solve kernels per step
    n = tpl.n_var
    flops_per_iter = n ** 3 / 3.0 + 2.0 * n ** 2          # dense-n LDL' + 2 solves
    achieved = (it_sum / n_meas) * flops_per_iter / (k_ms * 1e-3) / 1e12
    exec_flops = executed_flops_per_iter(tpl)
    cold_k = float(np.mean(cold_ms))
    traffic, traffic_source = measured_traffic(B)
    n_total = args.agents if strong else B * world
    out = {
        'metric': 'MPC solves/sec, 1024-agent Holonomic Point2point batch' + (' (strong scaling: the agents of ONE batch sharded over the GPUs)' if strong else ' per GPU'),
        'value': value, 'unit': 'solves/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
        'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'configs[1]: %d-agent Holonomic Point2point %s, degree 3, '
                               'knot_intervals=%d, %d circular obstacles; one step = one receding-horizon '
                               'MPC step of every agent (update_time 0.1 s, ideal prediction, primal-dual '
                               'warm start) after a cold solve; tol=%g'
                               % (n_total if strong else B, 'in total, sharded over the ranks' if strong else 'per GPU',
                                  args.knot_intervals, args.obstacles, args.tol),
                   'agents_per_gpu': B, 'agents_total': n_total, 'n_var': tpl.n_var, 'n_con': tpl.n_con,
                   'launches_per_step': n_parts,
                   'step_form': ('%d stream-ordered sub-batch launches per step (omgtools.batch.receding_horizon_batch: the batch is two '
                                 'rounds of resident workgroups)' % n_parts if n_parts > 1 else 'one launch per step'),
                   'parallelism': 'agents sharded across ranks, no collective on the solve path'},
        'tol': args.tol, 'compl_inf_tol': opts.get('compl_inf_tol', 0.0), 'constr_viol_tol': opts.get('constr_viol_tol', 0.0),
        'p50_batch_latency_ms': float(np.median(step_latency_ms)), 'max_batch_latency_ms': float(np.max(step_latency_ms)),
        'max_iters_in_a_step': int(stats[:, 2].max()), 'host_enqueue_ms_per_step': t_host / K * 1e3,
        'solved_fraction': n_ok / float(B), 'mean_iters': it_sum / float(n_meas * B),
        # co-headline: the cold solve of the whole batch from the reference's initial guess (SURVEY 8d target
        # >= 1e4 solves/s), with its own roofline object
        'cold_solve': {'solves_per_s': cold_ok / (cold_k * 1e-3), 'kernel_ms': cold_k,
                       'solved_fraction': cold_ok / float(B), 'mean_iters': cold_iters / float(B),
                       'first_pass': {'kernel_ms': float(np.mean(cold_first_ms)), 'solved_fraction': first_ok / float(B)},
                       'restart_passes': cold_passes,
                       'roofline': {'bound': 'mfma', 'kernel': 'ipm_solve_kernel',
                                    'achieved': cold_iters * flops_per_iter / (cold_k * 1e-3) / 1e12,
                                    'peak': FP64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                    'frac': cold_iters * flops_per_iter / (cold_k * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                                    'executed_TFLOPs': cold_iters * exec_flops / (cold_k * 1e-3) / 1e12,
                                    'traffic': None, 'kernel_ms': cold_k}},
        'lds_bytes_per_agent': solver.lds_bytes,
        'roofline': {'bound': 'mfma', 'kernel': 'ipm_solve_kernel', 'achieved': achieved,
                     'peak': FP64_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': achieved / FP64_MATRIX_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_source,
                     'kernel_ms': k_ms, 'launches_per_step': n_parts, 'launch_ms_mean': float(np.mean(all_ms[W * n_parts:])),
                     'executed_flops_per_iter': exec_flops, 'dense_n_flops_per_iter': flops_per_iter,
                     'executed_TFLOPs': (it_sum / n_meas) * exec_flops / (k_ms * 1e-3) / 1e12,
                     'frac_executed': (it_sum / n_meas) * exec_flops / (k_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                     'all_launches': {'n': len(launches_ms), 'mean_ms': float(np.mean(launches_ms)),
                                      'achieved': launches_iters * flops_per_iter / (sum(launches_ms) * 1e-3) / 1e12},
                     'note': 'algorithmic flops = sum(iters) x (n^3/3+2n^2), n=n_var (SURVEY 8d).  kernel_ms = chip time of the solve '
                             'kernels per timed step = UNION of the launch intervals (dispatch stamps of every launch, on the '
                             'stream it runs on) / steps: with sub-batch launches the launches of a step overlap each other and the '
                             'next step of the other sub-batches, so the sum of the launch durations (launch_ms_mean x launches) exceeds '
                             'it; all_launches = every launch of this process (cold solves + warm-up + timed steps, the set '
                             'rocprofv3 --stats averages: its mean_ms is per launch; its `achieved` divides by the SUM of the '
                             'durations and is a lower bound when launches overlap).  traffic: from the committed PMC passes '
                             '(traffic_source), not measured in this run'},
        'step_kernel_ms': [round(v, 3) for v in step_latency_ms], 'step_max_iters': [int(v) for v in stats[W:, 2]],
    }
    if isinstance(rh, StreamedP2P):
        rh.close()
    else:
        rh.solver.close()
    if world == 1 and not args.no_parity:
        # the accuracy the solver settings of `value` buy: closed loop against an independent solver in the loop (64 agents, 25 updates)
        try:
            out['parity_at_tol'] = closed_loop_parity(opts, dev)
            for k in ('closed_loop_pos_m', 'closed_loop_vel_mps', 'closed_loop_rel'):
                out[k] = out['parity_at_tol'][k]
        except Exception as e:
            out['parity_at_tol'] = {'error': repr(e)}
    if world == 1 and not args.no_extras:
        # the whole manoeuvre, and the accuracy / throughput curve over the solver settings (round-5 review, items 2 and 4)
        try:
            out['sustained'] = sustained_leg(problem, P, opts, args.sustained_steps, dev)
            out['sustained']['as_one_rollout'] = manoeuvre_rollout(problem, P, opts, args.sustained_steps, dev)
            every = sustained_leg(problem, P, opts, args.sustained_steps, dev, stop_rule=False)
            out['sustained']['every_agent_solved_at_every_update'] = {k: every[k] for k in ('solves_per_s', 'ms_per_update', 'mean_iters', 'max_iters', 'solves', 'note')}
        except Exception as e:
            out['sustained'] = {'error': repr(e)}
        from omgtools.backend import IPOPT_DEFAULT_TOLERANCES
        curve = []
        for label, o2 in (('tol 1e-3 + IPOPT default compl_inf_tol / constr_viol_tol 1e-4', dict(opts, tol=1e-3, **IPOPT_DEFAULT_TOLERANCES)),
                          ('tol 1e-4', dict(opts, tol=1e-4)), ('tol 1e-5', dict(opts, tol=1e-5)), ('tol 1e-6', dict(opts, tol=1e-6))):
            try:
                curve.append(dict(tolerance_leg(problem, P, o2, args.steps, args.warmup, dev), settings=label))
            except Exception as e:
                curve.append({'settings': label, 'error': repr(e)})
        out['tolerance_curve'] = curve
        out['latency_resident'] = latency_episodes(mpc, x0_init, p_init, 5, 40, host=False)
        out['latency_host_boundary'] = latency_episodes(mpc, x0_init, p_init, 5, 40, host=True)
        out['survey_8d_obstacle_rule'] = unedited_rule(args, dev, 20240807 + 2)
        out['trajectory_store_fused'] = store_leg(mpc, problem, tpl, x0_init, p_init, 20, dev)
        try:                                              # (the other form of the per-step path, for comparison)
            out['one_stream' if n_parts > 1 else 'two_streams'] = stepwise_leg(problem, P, opts, args.steps, args.warmup, dev, 1 if n_parts > 1 else 2)
        except Exception as e:
            out['one_stream' if n_parts > 1 else 'two_streams'] = {'error': repr(e)}
        for eng in ('mapped', 'kernel', 'memcpy'):
            key = 'host_boundary_pipelined' + ('' if eng == 'mapped' else '_' + eng)
            try:
                out[key] = host_boundary_pipelined(problem, P, opts, args.steps, args.warmup, dev, engine=eng)      # (the headline's own steps: later ones are harder)
            except Exception as e:
                out[key] = {'error': repr(e)}
        try:
            out['rollout'] = rollout_leg(mpc, x0_init, p_init, args.steps, args.warmup, dev)
        except Exception as e:                            # (a second metric must never cost the headline line)
            out['rollout'] = {'error': repr(e)}
        try:
            out['lifted_class'] = lifted_leg(256)
        except Exception as e:
            out['lifted_class'] = {'error': repr(e)}
    if world == 1:
        # trajectory extraction (A11) against the HBM roofline, at the workload's batch and at 16x (the
        # 49 MB of one 1024-agent launch last ~10 us: launch-latency bound)
        veh = problem.vehicles[0]
        out['roofline_sample'] = [sample_roofline(torch, tpl, veh, dev, nb, float(problem.options['horizon_time']))
                                  for nb in (B, 16 * B)]
    if not args.no_cpu and args.cpu_seconds > 0 and world == 1:
        out['cpu_baseline'] = cpu_baseline(problem, P, opts, args.steps, args.warmup, args.cpu_seconds)
    emit(out)


if __name__ == '__main__':
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()
