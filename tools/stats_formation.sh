# rocprofv3 per-kernel summary of the formation ADMM bench (evidence for profiles/r04_kernel_stats_formation.csv)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_form -- python $R/bench.py --workload formation --steps 100 --warmup 5 > $R/gpurun_out/prof_form_bench.json 2> $R/gpurun_out/prof_form.err )
f=$(find gpurun_out/prof_form -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = open('gpurun_out/r04_kernel_stats_formation.csv', 'w')
out.write('Name,Calls,TotalDurationNs,AverageNs,Percentage\n')
for r in rows[:12]:
    out.write('"%s",%s,%s,%s,%s\n' % (r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage']))
    print(r['Name'][:50], r['Calls'], r['AverageNs'], r['Percentage'])
PY
tail -c 200 gpurun_out/prof_form_bench.json
