#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
O=gpurun_out/q
AB=$GRAFT_REPO_ROOT/tools/scratch/ab
for v in r03 cur noprealloc keps100 ftb0 nosoc; do
  L=$AB/libomgx_$v.so; [ $v = cur ] && L=$GRAFT_REPO_ROOT/omg-tools_amd/csrc/libomgx.so
  OMGX_LIB=$L timeout 300 python bench.py --workload quadrotor --agents 4096 --steps 5 --warmup 2 > $O/quad_$v.json 2> $O/quad_$v.err
  OMGX_LIB=$L timeout 300 python bench.py --no-cpu --no-extras > $O/p2p_$v.json 2> $O/p2p_$v.err
  python - $O/quad_$v.json $O/p2p_$v.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); e = json.load(open(sys.argv[2]))
    rh = d.get('receding_horizon', {})
    print(sys.argv[1].split('/')[-1], 'cold %.0f/s %.1f ms iters %.1f' % (d['value'], d['ms_per_step'], d['mean_iters']), '| RH %.0f/s %.2f ms iters %.2f max %s' % (rh.get('solves_per_s', 0), rh.get('ms_per_step', 0), rh.get('mean_iters', 0), rh.get('max_iters')),
          '|| p2p %.0f cold %.0f' % (e['value'], e['cold_solve']['solves_per_s']))
except Exception as ex:
    print(sys.argv[1], 'FAILED', ex)
PY
done
