#!/bin/bash
# Round 5, GPU session 6: the lifted classes (Bicycle / AGV through the C ABI), then the whole GPU tier and the default bench
# (ABI 7: the template struct and the kernel's tables grew -- the headline instance must not have moved).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s6
mkdir -p $O
timeout 900 python -m pytest tests/test_lifted.py -m gpu -q -s > $O/pytest_lifted.log 2>&1; echo "pytest rc $?" >> $O/pytest_lifted.log
tail -n 15 $O/pytest_lifted.log
timeout 600 python bench.py --no-cpu --no-extras > $O/bench_quick.json 2> $O/bench_quick.err; echo "bench rc $?"
timeout 1500 python -m pytest tests -m gpu -q -s --deselect tests/test_lifted.py > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc |^FAILED" $O/pytest_gpu.log | tail -8
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
for n in ('bench_quick', 'bench_default'):
    try:
        e = json.load(open('gpurun_out/s6/%s.json' % n))
        print(n, '%.0f solves/s  %.3f ms/step  cold %.0f  frac %.4f' % (e['value'], e['ms_per_step'], e['cold_solve']['solves_per_s'], e['roofline']['frac']))
    except Exception as ex:
        print(n, 'unreadable', ex)
PY
