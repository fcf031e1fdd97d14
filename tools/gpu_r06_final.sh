#!/bin/bash
# final collection of round 6: GPU tier, the evidence set of tools/run_profiles.sh, the one-rank RCCL lines, the large batches
mkdir -p gpurun_out/final
python -m pytest tests -q -m gpu > gpurun_out/final/tests.log 2>&1; tail -n 4 gpurun_out/final/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -n 1 gpurun_out/final/smoke.log
bash tools/run_profiles.sh r06 > gpurun_out/final/run_profiles.log 2>&1
OMGX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --no-cpu --no-extras > gpurun_out/r06_bench_n1_torchrun_rccl_1rank.json 2> gpurun_out/final/rccl.err
OMGX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 1 --scaling strong --no-cpu --no-extras > gpurun_out/r06_bench_n1_strong_rccl_1rank.json 2>> gpurun_out/final/rccl.err
OMGX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29546 bench.py --gpus 1 --workload formation --steps 50 --warmup 5 > gpurun_out/r06_bench_formation_rccl_1rank.json 2>> gpurun_out/final/rccl.err
python bench.py --agents 8192 --no-cpu --no-extras > gpurun_out/r06_bench_n1_8192agents.json 2>/dev/null
python bench.py --agents 16384 --no-cpu --no-extras > gpurun_out/r06_bench_n1_16384agents.json 2>/dev/null
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value']), d['unit'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
P
