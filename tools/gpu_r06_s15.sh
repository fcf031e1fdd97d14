#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s15
for rep in 1 2; do
for cfg in "4 3" "5 3" "5 4" "6 3" "6 4" "6 5" "12 4" "12 6"; do set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --streams $2 --no-cpu --no-extras --no-parity > gpurun_out/s15/b_q$1_s$2_r$rep.json 2> gpurun_out/s15/b_q$1_s$2_r$rep.err
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/s15/b_*.json')):
    try:
        d = json.load(open(f)); print(f, 'value %.0f' % d['value'], 'p50 %.3f' % d['p50_batch_latency_ms'])
    except Exception as e: print(f, 'ERR', e)
PY
