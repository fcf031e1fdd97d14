#!/bin/bash
# Round-4 GPU session 1: parity tests of the new device paths, then same-box A/B of the headline against the round-3 library.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s1
O=gpurun_out/s1
( timeout 600 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "rc $?" >> $O/gputests.log )
B="python bench.py --no-cpu --no-extras"
for rep in 1 2; do
  OMGX_LIB=$GRAFT_REPO_ROOT/tools/scratch/ab/libomgx_r03.so timeout 300 $B > $O/bench_r03_$rep.json 2> $O/bench_r03_$rep.err
  OMGX_MAX_SOC=0 timeout 300 $B > $O/bench_new_soc0_$rep.json 2> $O/bench_new_soc0_$rep.err
  OMGX_MAX_SOC=1 timeout 300 $B > $O/bench_new_soc1_$rep.json 2> $O/bench_new_soc1_$rep.err
done
timeout 300 python tools/phase_profile.py 1024 > $O/phase_cold.json 2> $O/phase.err
timeout 300 python tools/phase_profile.py 1024 mpc > $O/phase_mpc.json 2>> $O/phase.err
OMGX_MAX_SOC=1 timeout 300 $B --agents 4096 > $O/bench_new_4096.json 2> $O/bench_new_4096.err
OMGX_MAX_SOC=1 timeout 300 $B --tol 1e-6 > $O/bench_new_tol6.json 2> $O/bench_new_tol6.err
timeout 300 python bench.py --workload formation --steps 50 --warmup 5 > $O/bench_formation.json 2> $O/bench_formation.err
timeout 300 python bench.py --workload rendezvous --steps 50 --warmup 5 > $O/bench_rendezvous.json 2> $O/bench_rendezvous.err
tail -3 $O/gputests.log
for f in $O/bench_r03_1 $O/bench_new_soc0_1 $O/bench_new_soc1_1 $O/bench_r03_2 $O/bench_new_soc0_2 $O/bench_new_soc1_2 $O/bench_new_4096 $O/bench_new_tol6 $O/bench_formation $O/bench_rendezvous; do python - $f.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'cold', d.get('cold_solve', {}).get('solves_per_s'), 'cold iters', d.get('cold_solve', {}).get('mean_iters'),
          'mean_iters', d.get('mean_iters', d.get('x_update_mean_iters')), 'max', d.get('step_max_iters', d.get('x_update_max_iters')), 'solved', d.get('solved_fraction'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
