#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s16
for rep in 1 2 3 4 5 6; do
OMGX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 1 --no-cpu --no-extras --no-parity > gpurun_out/s16/q8_r$rep.json 2> gpurun_out/s16/q8_r$rep.err
OMGX_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 1 --no-cpu --no-extras --no-parity > gpurun_out/s16/q4_r$rep.json 2> gpurun_out/s16/q4_r$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/s16/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.0f' % d['value'], 'p50 %.3f' % d['p50_batch_latency_ms'], d['config']['launches_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
