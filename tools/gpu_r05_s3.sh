#!/bin/bash
# Round 5, GPU session 3: the whole GPU test tier (no -x), default bench.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s3
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
grep -E "closed loop|passed|failed|rc |^FAILED" $O/pytest_gpu.log | tail -12
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
e = json.load(open('gpurun_out/s3/bench_default.json'))
print('%.0f solves/s  %.3f ms/step  p50 %.3f  kernel %.3f  frac %.4f exec %.3f  cold %.0f (%.1f it) enq %.3f' % (e['value'], e['ms_per_step'], e['p50_batch_latency_ms'], e['roofline']['kernel_ms'], e['roofline']['frac'], e['roofline']['executed_TFLOPs'], e['cold_solve']['solves_per_s'], e['cold_solve']['mean_iters'], e['host_enqueue_ms_per_step']), e['step_max_iters'])
for k in ('one_stream', 'rollout', 'host_boundary_pipelined', 'host_boundary_pipelined_memcpy', 'latency_host_boundary'):
    if k in e: print('   ', k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e[k].items() if a not in ('note', 'step_max_iters')})
PY
