#!/bin/bash
# Round 5, GPU session 5: GPU test tier (new closed-loop cases), host-boundary probe, default bench.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s5
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
grep -E "closed loop|warm steps|passed|failed|rc |^FAILED" $O/pytest_gpu.log | tail -14
timeout 300 python tools/hb_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/hb_probe.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
e = json.load(open('gpurun_out/s5/bench_default.json'))
print('%.0f solves/s  %.3f ms/step  cold %.0f' % (e['value'], e['ms_per_step'], e['cold_solve']['solves_per_s']))
for k in ('host_boundary_pipelined', 'host_boundary_pipelined_kernel', 'host_boundary_pipelined_memcpy'):
    if k in e: print('   ', k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e[k].items() if a not in ('note',)})
PY
