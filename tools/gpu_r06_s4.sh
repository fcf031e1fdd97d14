#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
( timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s4/gputests.log 2>&1; echo "rc $?" >> gpurun_out/s4/gputests.log )
tail -8 gpurun_out/s4/gputests.log
timeout 900 python bench.py > gpurun_out/s4/bench.json 2> gpurun_out/s4/bench.err; echo "bench rc $?"
timeout 300 python bench.py --ipopt-defaults --no-cpu --no-extras > gpurun_out/s4/bench_ipopt_defaults.json 2> gpurun_out/s4/bench_ipd.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/s4/bench.json'))
print('value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'cold', d['cold_solve']['solves_per_s'], 'p50', d['p50_batch_latency_ms'])
print('sustained', {k: v for k, v in d.get('sustained', {}).items() if k not in ('windows', 'note')})
for c in d.get('tolerance_curve', []):
    print('curve', c.get('settings'), 'per-step %.0f' % c.get('solves_per_s', 0), 'rollout', c.get('rollout'), c.get('mean_iters'), c.get('max_iters_in_a_step'), {k: c.get('parity', {}).get(k) for k in ('closed_loop_pos_m', 'closed_loop_rel')}, c.get('error'))
e = json.load(open('gpurun_out/s4/bench_ipopt_defaults.json'))
print('ipopt-defaults line: value %.0f' % e['value'], e['mean_iters'], e['max_iters_in_a_step'], e.get('closed_loop_pos_m'))
PY
