#!/bin/bash
# sub-batch count on high-priority streams (their own pool of four hardware queues), plain process
mkdir -p gpurun_out/s25
for rep in 1 2; do for s in 2 3 4 5; do
python bench.py --no-cpu --no-extras --no-parity --streams $s > gpurun_out/s25/s${s}_$rep.json 2>/dev/null
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/s25/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), d['ms_per_step'], d.get('p50_batch_latency_ms'), d.get('latency_ms', {}) if isinstance(d.get('latency_ms'), dict) else '')
P
