#!/bin/bash
# Round 5, GPU session 2: test tier with the Gershgorin clamp / transfer kernel, default bench, stream-count sweep.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s2
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
grep -E "closed loop|passed|failed|rc " $O/pytest_gpu.log | tail -8
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
for n in 1 2 4; do
  timeout 300 python bench.py --streams $n --no-cpu --no-extras > $O/bench_streams$n.json 2> $O/bench_streams$n.err
done
timeout 300 python bench.py --agents 4096 --no-cpu --no-extras > $O/bench_4096.json 2> $O/bench_4096.err
python - <<'PY'
import json
for n in ('bench_default', 'bench_streams1', 'bench_streams2', 'bench_streams4', 'bench_4096'):
    try:
        e = json.load(open('gpurun_out/s2/%s.json' % n))
        print(n, '%.0f solves/s  %.3f ms/step  p50 %.3f  kernel %.3f  frac %.4f exec %.3f  cold %.0f (%.1f it) enq %.3f' % (e['value'], e['ms_per_step'], e['p50_batch_latency_ms'], e['roofline']['kernel_ms'], e['roofline']['frac'], e['roofline']['executed_TFLOPs'], e['cold_solve']['solves_per_s'], e['cold_solve']['mean_iters'], e['host_enqueue_ms_per_step']), e['step_max_iters'])
        for k in ('one_stream', 'two_streams', 'rollout', 'host_boundary_pipelined', 'host_boundary_pipelined_memcpy', 'latency_host_boundary'):
            if k in e: print('   ', k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e[k].items() if a not in ('note', 'step_max_iters')})
    except Exception as ex:
        print(n, 'FAILED', ex)
PY
