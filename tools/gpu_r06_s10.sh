#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s10
for rep in 1 2; do
for cfg in "4 3" "8 3" "8 4" "8 5" "8 6" "16 8"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --streams $2 --no-cpu --no-extras --no-parity > gpurun_out/s10/b_q$1_s$2_r$rep.json 2> gpurun_out/s10/b_q$1_s$2_r$rep.err
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/s10/b_*.json')):
    try:
        d = json.load(open(f)); print(f, 'value %.0f' % d['value'], 'p50 %.3f' % d['p50_batch_latency_ms'], 'max %.3f' % d['max_batch_latency_ms'], d['config']['launches_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
