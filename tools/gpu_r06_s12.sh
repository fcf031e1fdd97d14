#!/bin/bash
# final checks of round 6: GPU tier, smoke(), the RCCL path with one rank (weak + strong + formation), the default bench once more
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s12
( timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s12/gputests.log 2>&1; echo "rc $?" >> gpurun_out/s12/gputests.log )
tail -4 gpurun_out/s12/gputests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s12/smoke.log 2>&1; tail -2 gpurun_out/s12/smoke.log
OMGX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --no-cpu --no-extras > gpurun_out/r06_bench_n1_torchrun_rccl_1rank.json 2> gpurun_out/s12/rccl.err
OMGX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 1 --scaling strong --no-cpu --no-extras > gpurun_out/r06_bench_n1_strong_rccl_1rank.json 2>> gpurun_out/s12/rccl.err
OMGX_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29546 bench.py --gpus 1 --workload formation --steps 50 --warmup 5 > gpurun_out/r06_bench_formation_rccl_1rank.json 2>> gpurun_out/s12/rccl.err
( time python bench.py > gpurun_out/s12/bench_default.json 2> gpurun_out/s12/bench_default.err ) 2> gpurun_out/s12/bench_time.txt
tail -3 gpurun_out/s12/bench_time.txt
python - <<'PY'
import json
for f in ('gpurun_out/r06_bench_n1_torchrun_rccl_1rank.json', 'gpurun_out/r06_bench_n1_strong_rccl_1rank.json', 'gpurun_out/r06_bench_formation_rccl_1rank.json', 'gpurun_out/s12/bench_default.json'):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['metric'][:40], '%.0f' % d['value'], d.get('n_gpus'), d.get('closed_loop_pos_m'))
    except Exception as e: print(f, 'ERR', e)
PY
