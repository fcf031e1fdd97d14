#!/bin/bash
# same-box A/B of library builds on the default bench: bash tools/gpu_ab.sh name1 name2 ... (tools/scratch/ab/libomgx_<name>.so; "cur" = the tree's)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
AB=$GRAFT_REPO_ROOT/tools/scratch/ab
for rep in 1 2; do
for v in "$@"; do
  L=$AB/libomgx_$v.so; [ $v = cur ] && L=$GRAFT_REPO_ROOT/omg-tools_amd/csrc/libomgx.so
  OMGX_LIB=$L timeout 300 python bench.py --no-cpu --no-extras > gpurun_out/ab/p2p_${v}_$rep.json 2> gpurun_out/ab/p2p_${v}_$rep.err
  python - gpurun_out/ab/p2p_${v}_$rep.json <<'PY'
import json, sys
try:
    e = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], 'p2p %.0f  p50 %.3f ms  cold %.0f (%.1f it)' % (e['value'], e['p50_batch_latency_ms'], e['cold_solve']['solves_per_s'], e['cold_solve']['mean_iters']), e['step_max_iters'])
except Exception as ex:
    print(sys.argv[1], 'FAILED', ex)
PY
done
done
