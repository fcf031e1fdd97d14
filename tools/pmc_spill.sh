# PMC passes for the spill classes (cold solve of a 1024-agent batch each; separate runs, counters never combined with trace domains):
# HBM traffic and matrix-pipe / wave-wait shares -> gpurun_out/${TAG:-r06}_pmc_spill_<class>.txt
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cls in quadrotor_p2p holonomic3d_p2p; do
  out=$R/gpurun_out/${TAG:-r06}_pmc_spill_$cls.txt; : > $out
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
    d=$R/gpurun_out/prof_spill/$cls/$(echo $set | tr ' ' '_')
    mkdir -p $R/gpurun_out/prof_spill/$cls; ( cd /tmp && timeout 120 rocprofv3 --pmc $set --output-format csv -d $d -- python $R/tools/cold_hist.py $cls > $d.log 2>&1 )
    for f in $(find $d -name "*counter_collection.csv"); do python - "$f" >> $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = {}
for r in rows:
    if 'ipm_solve' in r['Kernel_Name']:
        acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, v in acc.items():
    print(k, 'launches', len(v), 'per launch', [round(x) for x in v][:4])
PY
    done
  done
  grep -h "first pass" $R/gpurun_out/prof_spill/$cls/FETCH_SIZE.log >> $out
  echo "== $cls"; cat $out
done
