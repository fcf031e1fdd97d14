#!/bin/bash
# Round 5, GPU session 1: GPU test tier, the default bench (two half-launches per step), the single-handle form next to it,
# and the instruction-fetch / issue PMC passes the round-4 verdict asked for.  gpurun -- 'bash tools/gpu_r05_s1.sh'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s1
mkdir -p $O $R/gpurun_out/prof
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
timeout 300 python bench.py --streams 1 --no-cpu --no-extras > $O/bench_one_stream.json 2> $O/bench_one_stream.err
timeout 300 python bench.py --no-cpu --no-extras > $O/bench_two_streams.json 2> $O/bench_two_streams.err
timeout 300 python bench.py --agents 256 --no-cpu --no-extras > $O/bench_256.json 2> $O/bench_256.err
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ|SQC|TCP|TCC|GRBM)_[A-Z0-9_]+" | sort -u > $O/counters_available.txt
B="python $R/bench.py --streams 1 --no-cpu --no-extras --steps 5 --warmup 1"
pass() {  # name, counters...
  n=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/prof/$n -- $B > /dev/null 2> $R/gpurun_out/prof/$n.err ); echo "pmc $n rc $?"
}
pass issue1 SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_IFETCH
pass issue2 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES
pass issue3 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_WAVE_CYCLES
python tools/profile_summary.py r05 issue > $O/issue_summary.txt 2>&1
cat $O/issue_summary.txt | tail -40
python - <<'PY'
import json
for n in ('bench_default', 'bench_one_stream', 'bench_two_streams', 'bench_256'):
    try:
        e = json.load(open('gpurun_out/s1/%s.json' % n))
        print(n, '%.0f solves/s  %.3f ms/step  p50 %.3f  kernel %.3f  frac %.4f exec %.3f  cold %.0f (%.1f it)' % (e['value'], e['ms_per_step'], e['p50_batch_latency_ms'], e['roofline']['kernel_ms'], e['roofline']['frac'], e['roofline']['executed_TFLOPs'], e['cold_solve']['solves_per_s'], e['cold_solve']['mean_iters']), e['step_max_iters'])
        for k in ('one_stream', 'two_streams', 'rollout', 'host_boundary_pipelined', 'latency_host_boundary'):
            if k in e: print('   ', k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e[k].items() if a not in ('note', 'step_max_iters')})
        if 'cpu_baseline' in e: print('    cpu', e['cpu_baseline']['value'], e['cpu_baseline'].get('slsqp_restatement'))
    except Exception as ex:
        print(n, 'FAILED', ex)
PY
