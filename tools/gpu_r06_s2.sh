#!/bin/bash
# round 6, step 2: where does the time go with the setup kernel on / off -- phase cycles of the solve kernel (profiling build) and
# rocprofv3 kernel durations of the one-launch-per-step bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/s2
R=$GRAFT_REPO_ROOT
for on in 1 0; do
  OMGX_PREPARE=$on python tools/phase_profile.py 1024 mpc > gpurun_out/s2/phase_mpc_p${on}.json 2> gpurun_out/s2/phase_p${on}.err
  ( cd /tmp && OMGX_PREPARE=$on rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s2/stats_p${on} -- python $R/bench.py --streams 1 --no-cpu --no-extras > $R/gpurun_out/s2/bench_p${on}.json 2> $R/gpurun_out/s2/stats_p${on}.err )
done
python - <<'PY'
import json, glob, csv
for on in (1, 0):
    d = json.load(open('gpurun_out/s2/phase_mpc_p%d.json' % on))
    c = d['cycles_per_solve']
    print('prepare', on, 'kernel_ms', d['kernel_ms_p50'], 'total', c['total'], 'setup', c['setup'], {k: round(c[k]) for k in ('s_params','s_jac0','s_class','s_init','jac','resid','assemble','factor','solve','step','linesearch','update')})
    for f in glob.glob('gpurun_out/s2/stats_p%d/**/*kernel_stats.csv' % on, recursive=True):
        for row in csv.DictReader(open(f)):
            if 'ipm_' in row['Name'] or 'predict' in row['Name']:
                print('   ', row['Name'][:60], row['Calls'], row['AverageNs'], row['MinNs'], row['MaxNs'])
PY
