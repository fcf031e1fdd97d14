#!/bin/bash
# round 6: second-order correction at every step length a row rejects (wave-path templates) -- GPU tier, full bench line, phase cycles, A/B lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s17
( timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s17/gputests.log 2>&1; echo "rc $?" >> gpurun_out/s17/gputests.log )
tail -8 gpurun_out/s17/gputests.log
timeout 900 python bench.py > gpurun_out/s17/bench.json 2> gpurun_out/s17/bench.err; echo "bench rc $?"
for rep in 1 2; do
timeout 300 python bench.py --streams 1 --no-cpu --no-extras --no-parity > gpurun_out/s17/bench1_r$rep.json 2>/dev/null
timeout 300 python bench.py --ipopt-defaults --no-cpu --no-extras > gpurun_out/s17/bench_ipd_r$rep.json 2>/dev/null
done
timeout 300 python bench.py --tol 1e-6 --no-cpu --no-extras > gpurun_out/s17/bench_tol1e-6.json 2>/dev/null
python tools/phase_profile.py 1024 mpc > gpurun_out/s17/phase_mpc.json 2> gpurun_out/s17/phase.err
python tools/phase_profile.py 1024 > gpurun_out/s17/phase_cold.json 2>> gpurun_out/s17/phase.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/s17/bench.json'))
print('value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'cold %.0f iters %.2f' % (d['cold_solve']['solves_per_s'], d['cold_solve']['mean_iters']), 'p50', d['p50_batch_latency_ms'], 'maxit', d['max_iters_in_a_step'])
print('sustained', {k: v for k, v in d.get('sustained', {}).items() if k in ('solves_per_s', 'all_agents_solves_per_s', 'mean_iters', 'max_iters')})
for c in d.get('tolerance_curve', []):
    print('curve', c.get('settings'), 'per-step %.0f' % c.get('solves_per_s', 0), 'rollout %.0f' % c.get('rollout', {}).get('solves_per_s', 0), c.get('mean_iters'), c.get('max_iters_in_a_step'), c.get('solved_fraction'), {k: c.get('parity', {}).get(k) for k in ('closed_loop_pos_m', 'closed_loop_rel')})
for f in ('bench1_r1', 'bench1_r2', 'bench_ipd_r1', 'bench_ipd_r2', 'bench_tol1e-6'):
    e = json.load(open('gpurun_out/s17/%s.json' % f)); print(f, '%.0f' % e['value'], e['mean_iters'], e['max_iters_in_a_step'], e['solved_fraction'], 'cold %.0f' % e['cold_solve']['solves_per_s'])
c = json.load(open('gpurun_out/s17/phase_mpc.json'))['cycles_per_solve']; print('mpc total %.0f linesearch %.0f' % (c['total'], c['linesearch']))
PY
