#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s14
for rep in 1 2; do
OMGX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 1 --no-cpu --no-extras > gpurun_out/s14/dist_r$rep.json 2> gpurun_out/s14/dist_r$rep.err
OMGX_FORCE_DIST=1 GPU_MAX_HW_QUEUES=4 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 1 --no-cpu --no-extras > gpurun_out/s14/dist_q4_r$rep.json 2> gpurun_out/s14/dist_q4_r$rep.err
python bench.py --no-cpu --no-extras > gpurun_out/s14/plain_r$rep.json 2> gpurun_out/s14/plain_r$rep.err
done
OMGX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --scaling strong --no-cpu --no-extras > gpurun_out/r06_bench_n1_strong_rccl_1rank.json 2>> gpurun_out/s14/rccl.err
OMGX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29546 bench.py --gpus 1 --workload formation --steps 50 --warmup 5 > gpurun_out/r06_bench_formation_rccl_1rank.json 2>> gpurun_out/s14/rccl.err
cp gpurun_out/s14/dist_r1.json gpurun_out/r06_bench_n1_torchrun_rccl_1rank.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/s14/*.json')) + ['gpurun_out/r06_bench_n1_strong_rccl_1rank.json', 'gpurun_out/r06_bench_formation_rccl_1rank.json']:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.0f' % d['value'], 'p50 %s' % d.get('p50_batch_latency_ms'), d['config'].get('launches_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
