#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
O=gpurun_out/s3
( timeout 1200 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "rc $?" >> $O/gputests.log )
tail -12 $O/gputests.log
B="python bench.py --no-cpu --no-extras"
for rep in 1 2 3; do
  OMGX_ORDER_DW=0 timeout 300 $B > $O/bench_order0_$rep.json 2> $O/bench_order0_$rep.err
  OMGX_ORDER_DW=1 timeout 300 $B > $O/bench_order1_$rep.json 2> $O/bench_order1_$rep.err
done
for f in $O/bench_order0_1 $O/bench_order1_1 $O/bench_order0_2 $O/bench_order1_2 $O/bench_order0_3 $O/bench_order1_3; do python - $f.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'p50', d.get('p50_batch_latency_ms'), 'cold', d['cold_solve']['solves_per_s'], [round(v, 2) for v in d['step_kernel_ms']])
PY
done
