# extra counter passes (each on its own, never combined with trace domains): LDS conflicts and wave stall split
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
( cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/prof/lds -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > /dev/null 2> $R/gpurun_out/prof/lds.err )
( cd /tmp && rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/prof/wait -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > /dev/null 2> $R/gpurun_out/prof/wait.err )
ls gpurun_out/prof/lds/* gpurun_out/prof/wait/* | head
