#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s7
( timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s7/gputests.log 2>&1; echo "rc $?" >> gpurun_out/s7/gputests.log )
tail -8 gpurun_out/s7/gputests.log
python bench.py --workload holonomic3d --agents 8192 --steps 3 --warmup 1 > gpurun_out/s7/bench_h3d8192.json 2> gpurun_out/s7/h3d.err
python bench.py --workload holonomic3d --steps 3 --warmup 1 > gpurun_out/s7/bench_h3d.json 2>> gpurun_out/s7/h3d.err
python - <<'PY'
import json
for f in ('gpurun_out/s7/bench_h3d8192.json', 'gpurun_out/s7/bench_h3d.json'):
    d = json.load(open(f))
    print(f, d.get('metric'), d.get('value'), {k: d.get(k) for k in ('cold_solve', 'mean_iters') if k in d})
PY
