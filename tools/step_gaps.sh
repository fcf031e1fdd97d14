# GPU timeline of the receding-horizon steps: kernel start/end stamps of a short bench run, gaps between launches
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace -- python $R/bench.py --no-cpu --no-extras --steps 12 --warmup 3 $OMGX_BENCH_ARGS > $R/gpurun_out/trace_bench.json 2> $R/gpurun_out/trace.err )
python - <<'PY'
import csv, glob, os
root = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/trace'
ev = []
for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]))
for f in glob.glob(root + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
ev.sort()
# the last 12 solve kernels = the timed steps
idx = [i for i, e in enumerate(ev) if 'ipm_solve' in e[2]]
first = idx[-6]
prev_end = None
for s, e, n in ev[first - 2:idx[-1] + 3]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print('%-42s dur %8.1f us   gap before %7.1f us' % (n, (e - s) / 1e3, gap))
    prev_end = e
PY
