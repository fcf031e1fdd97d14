# HBM traffic of one cold solve of the config-5 class (1024 agents): separate PMC passes (never with trace domains)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_h3d/fetch -- python $R/tools/cold_hist.py ${1:-holonomic3d_p2p} > $R/gpurun_out/prof_h3d_fetch.log 2>&1 )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_h3d/write -- python $R/tools/cold_hist.py ${1:-holonomic3d_p2p} > $R/gpurun_out/prof_h3d_write.log 2>&1 )
grep "first pass" gpurun_out/prof_h3d_fetch.log gpurun_out/prof_h3d_write.log
for f in $(find gpurun_out/prof_h3d -name "*counter_collection.csv"); do echo $f; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = {}
for r in rows:
    if 'ipm_solve' in r['Kernel_Name']:
        acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k, 'launches', len(v), 'sum per launch', [round(x) for x in v][:4])
PY
done
