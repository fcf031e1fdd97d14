#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
O=gpurun_out/s2
( timeout 900 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "rc $?" >> $O/gputests.log )
tail -15 $O/gputests.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload formation --steps 50 --warmup 5 > $O/bench_formation.json 2> $O/bench_formation.err
timeout 300 python bench.py --workload rendezvous --steps 50 --warmup 5 > $O/bench_rendezvous.json 2> $O/bench_rendezvous.err
timeout 300 python bench.py --workload quadrotor --agents 4096 --steps 5 --warmup 2 > $O/bench_quadrotor_4096.json 2> $O/bench_quadrotor_4096.err
timeout 300 python bench.py --workload holonomic3d --agents 8192 --steps 3 --warmup 1 > $O/bench_h3d_8192.json 2> $O/bench_h3d_8192.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for f in bench_default bench_formation bench_rendezvous bench_quadrotor_4096 bench_h3d_8192; do python - $O/$f.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'cold', d.get('cold_solve', {}).get('solves_per_s'),
          'iters', d.get('mean_iters', d.get('x_update_mean_iters')), 'max', d.get('max_iters_in_a_step', d.get('x_update_max_iters')), 'solved', d.get('solved_fraction'),
          'cpu', d.get('cpu_baseline', {}).get('value'), 'phases', d.get('phase_ms'), 'frac', d.get('roofline', {}).get('frac'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(sys.argv[1].replace('.json', '.err')).read()[-1500:])
PY
done
