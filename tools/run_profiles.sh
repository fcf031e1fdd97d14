cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -- python $R/bench.py --no-cpu > $R/gpurun_out/prof/bench_stats.json 2> $R/gpurun_out/prof/stats.err )
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/fetch -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > /dev/null 2> $R/gpurun_out/prof/fetch.err )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/write -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > /dev/null 2> $R/gpurun_out/prof/write.err )
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $R/gpurun_out/prof/mfma -- python $R/bench.py --steps 5 --warmup 1 --no-cpu > /dev/null 2> $R/gpurun_out/prof/mfma.err )
python bench.py --workload formation --no-cpu > gpurun_out/bench_formation.json 2> gpurun_out/bench_formation.err
python bench.py --workload quadrotor --agents 1024 --steps 2 --warmup 1 --no-cpu > gpurun_out/bench_quadrotor.json 2> gpurun_out/bench_quadrotor.err
python bench.py --workload holonomic3d --agents 1024 --steps 2 --warmup 1 --no-cpu > gpurun_out/bench_holonomic3d.json 2> gpurun_out/bench_holonomic3d.err
ls -R gpurun_out/prof | head -40
python tools/phase_profile.py 1024 > gpurun_out/phase_cold_final.json 2> gpurun_out/phase_final.err
python tools/phase_profile.py 1024 warm > gpurun_out/phase_warm_final.json 2>> gpurun_out/phase_final.err
./tools/micro/lat > gpurun_out/micro_latency.txt 2>&1
./tools/micro/ldl > gpurun_out/micro_ldl.txt 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
