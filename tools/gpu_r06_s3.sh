#!/bin/bash
# round 6, step 3: the GPU tier (every failure listed), then the full default bench line (parity_at_tol, tolerance curve, sustained leg)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
( timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s3/gputests.log 2>&1; echo "rc $?" >> gpurun_out/s3/gputests.log )
tail -15 gpurun_out/s3/gputests.log
timeout 900 python bench.py > gpurun_out/s3/bench.json 2> gpurun_out/s3/bench.err; echo "bench rc $?"
tail -3 gpurun_out/s3/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/s3/bench.json'))
print('value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'cold', d['cold_solve']['solves_per_s'], 'p50', d['p50_batch_latency_ms'])
print('parity', {k: d['parity_at_tol'].get(k) for k in ('tol', 'closed_loop_pos_m', 'closed_loop_vel_mps', 'closed_loop_rel', 'error')})
print('sustained', {k: v for k, v in d.get('sustained', {}).items() if k not in ('windows', 'note')})
for w in d.get('sustained', {}).get('windows', []):
    print('   ', w)
for c in d.get('tolerance_curve', []):
    print('curve', c.get('settings'), c.get('solves_per_s'), c.get('mean_iters'), c.get('solved_fraction'), c.get('max_iters_in_a_step'), {k: c.get('parity', {}).get(k) for k in ('closed_loop_pos_m', 'closed_loop_rel', 'solves_at_iteration_cap')}, c.get('error'))
print('one_stream', d.get('one_stream', {}).get('solves_per_s'), 'rollout', d.get('rollout', {}).get('solves_per_s'))
PY
