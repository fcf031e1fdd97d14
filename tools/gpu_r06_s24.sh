#!/bin/bash
# under a process group with the runtime's default four hardware queues: do three high-priority sub-batch streams keep their overlap?
mkdir -p gpurun_out/s24
for s in 3 2 4; do
OMGX_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4 python bench.py --no-cpu --no-extras --no-parity --streams $s > gpurun_out/s24/q4_s$s.json 2>/dev/null
OMGX_FORCE_DIST=1 python bench.py --no-cpu --no-extras --no-parity --streams $s > gpurun_out/s24/q8_s$s.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/s24/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), d['ms_per_step'])
P
