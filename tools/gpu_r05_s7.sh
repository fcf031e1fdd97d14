#!/bin/bash
# Round 5, GPU session 7 (final library, ABI 7): the GPU tier, the lifted-class probe, then the evidence under profiles/ again.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/s7
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
grep -E "passed|failed|rc |^FAILED" $O/pytest_gpu.log | tail -8
for n in 13 512 2048; do timeout 600 python tools/lifted_probe.py $n 2> /dev/null | tail -1 >> $O/r05_lifted_agv_probe.txt; done
cat $O/r05_lifted_agv_probe.txt | cut -c1-400
bash tools/run_profiles.sh r05 > $O/run_profiles.log 2>&1
tail -12 $O/run_profiles.log
python - <<'PY'
import json
e = json.load(open('gpurun_out/r05_bench_n1.json'))
print('%.0f solves/s  %.3f ms/step  cold %.0f  frac %.4f' % (e['value'], e['ms_per_step'], e['cold_solve']['solves_per_s'], e['roofline']['frac']))
PY
