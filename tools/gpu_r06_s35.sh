#!/bin/bash
# re-tune of the two scheduling knobs of the solve kernel on the high-priority streams: start stagger of the second workgroup of a CU, wave priority of long solves
mkdir -p gpurun_out/s35
for rep in 1 2; do
for v in "def" "OMGX_STAGGER=0" "OMGX_STAGGER=2" "OMGX_STAGGER=8" "OMGX_PRIO_ITER=0" "OMGX_PRIO_ITER=2" "OMGX_PRIO_ITER=6"; do
  if [ "$v" = def ]; then python bench.py --no-cpu --no-extras --no-parity > gpurun_out/s35/x.json 2>/dev/null; else env $v python bench.py --no-cpu --no-extras --no-parity > gpurun_out/s35/x.json 2>/dev/null; fi
  python - "$v" <<'P'
import json,sys
d=json.loads(open('gpurun_out/s35/x.json').read().strip().splitlines()[-1]); print('%-18s %9d solves/s  cold %.0f' % (sys.argv[1], round(d['value']), d['cold_solve']['solves_per_s']), flush=True)
P
done; done 2>&1 | tee gpurun_out/s35.txt
