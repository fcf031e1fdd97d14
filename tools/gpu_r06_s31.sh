#!/bin/bash
mkdir -p gpurun_out/s31
python -m pytest tests/test_gpu_rollout.py tests/test_gpu_batch_mpc.py tests/test_gpu_perf_guard.py -q -m gpu 2>&1 | tail -n 12
python bench.py --no-cpu > gpurun_out/s31/full.json 2> gpurun_out/s31/full.err; tail -n 2 gpurun_out/s31/full.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/s31/full.json').read().strip().splitlines()[-1]); s=d['sustained']
print(round(d['value']), 'sustained', s.get('solves_per_s'), 'rollout', s.get('as_one_rollout'), 'plain rollout', d['rollout']['solves_per_s'])
P
