#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s13
run() { # name, env, extra args
  env $2 OMGX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 1 --no-cpu --no-extras --no-parity $3 > gpurun_out/s13/$1.json 2> gpurun_out/s13/$1.err
}
for rep in 1 2; do
run s3_r$rep "A=1" "--streams 3"
run s2_r$rep "A=1" "--streams 2"
run s1_r$rep "A=1" "--streams 1"
run s3q8_r$rep "GPU_MAX_HW_QUEUES=8" "--streams 3"
run s4q8_r$rep "GPU_MAX_HW_QUEUES=8" "--streams 4"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/s13/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.0f' % d['value'], 'p50 %.3f' % d['p50_batch_latency_ms'], d['config']['launches_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
