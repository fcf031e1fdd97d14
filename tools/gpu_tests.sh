#!/bin/bash
# GPU tier of the test suite on the box (no -x: every failure is listed), then the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t
( timeout 900 python -m pytest tests -m gpu -q > gpurun_out/t/gputests.log 2>&1; echo "rc $?" >> gpurun_out/t/gputests.log )
tail -25 gpurun_out/t/gputests.log
timeout 300 python bench.py --no-cpu --no-extras > gpurun_out/t/bench.json 2> gpurun_out/t/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/t/bench.json'))
print('value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'cold', d['cold_solve']['solves_per_s'], d['cold_solve']['mean_iters'], 'max iters', d['step_max_iters'])
PY
