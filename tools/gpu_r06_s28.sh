#!/bin/bash
mkdir -p gpurun_out/s28
for rep in 1 2; do python bench.py --no-cpu > gpurun_out/s28/full_$rep.json 2>/dev/null; done
python bench.py > gpurun_out/s28/full_3.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/s28/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('rollout') or {}; s=d.get('sustained') or {}
    print(f.split('/')[-1], round(d['value']), 'rollout', r.get('solves_per_s'), r.get('kernel_ms_per_step'), 'sustained', s.get('solves_per_s'), 'curve rollouts', [round((c.get('rollout') or {}).get('solves_per_s',0)) for c in d.get('tolerance_curve',[])], 'curve', [round(c.get('solves_per_s',0)) for c in d.get('tolerance_curve',[])])
P
