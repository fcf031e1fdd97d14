#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s9
for rep in 1 2; do
for cfg in "3 0" "4 1" "4 0" "3 1" "5 1"; do set -- $cfg
  OMGX_EXP_CALLER_STREAM=$2 timeout 300 python bench.py --streams $1 --no-cpu --no-extras --no-parity > gpurun_out/s9/b_$1_$2_r$rep.json 2> gpurun_out/s9/b_$1_$2_r$rep.err
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/s9/b_*.json')):
    try:
        d = json.load(open(f)); print(f, 'value %.0f' % d['value'], 'p50 %.3f' % d['p50_batch_latency_ms'], d['config']['launches_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
