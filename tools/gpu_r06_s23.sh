#!/bin/bash
# sub-batch streams shared per device, high priority (their own hardware-queue pool): bench plain / under a process group, tests
mkdir -p gpurun_out/s23
python bench.py > gpurun_out/s23/bench.json 2> gpurun_out/s23/bench.err; tail -n 2 gpurun_out/s23/bench.err
OMGX_FORCE_DIST=1 python bench.py --no-cpu > gpurun_out/s23/bench_rccl1.json 2> gpurun_out/s23/bench_rccl1.err; tail -n 2 gpurun_out/s23/bench_rccl1.err
OMGX_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4 python bench.py --no-cpu --no-extras --no-parity > gpurun_out/s23/bench_rccl1_q4.json 2> gpurun_out/s23/bench_rccl1_q4.err
python bench.py --no-cpu --no-extras --no-parity --streams 1 > gpurun_out/s23/bench_1launch.json 2> gpurun_out/s23/bench_1launch.err
python - <<'P'
import json
for f in ('bench','bench_rccl1','bench_rccl1_q4','bench_1launch'):
    try:
        d=json.loads(open('gpurun_out/s23/%s.json'%f).read().strip().splitlines()[-1])
        print(f, round(d['value']), d['ms_per_step'], d.get('launches_per_step'), (d.get('sustained') or {}).get('solves_per_s'), (d.get('sustained') or {}).get('ms_per_update'))
        for c in d.get('tolerance_curve', []): print('   ', c.get('settings'), round(c.get('solves_per_s', 0)), c.get('rollout', {}).get('solves_per_s') if isinstance(c.get('rollout'), dict) else None)
    except Exception as e: print(f, 'ERR', e)
P
python -m pytest tests -x -q -m gpu > gpurun_out/s23/tests.log 2>&1; tail -n 3 gpurun_out/s23/tests.log
