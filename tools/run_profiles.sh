# One collection run of the evidence under profiles/ (round tag = $1, default r03).  Run on the GPU box:
#   gpurun -- 'bash tools/run_profiles.sh r03'      then   python tools/profile_summary.py r03
# The stats pass and every PMC pass are separate rocprofv3 runs (counters are never combined with trace domains).
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu --no-extras"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -- $B > $R/gpurun_out/prof/bench_stats.json 2> $R/gpurun_out/prof/stats.err )
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/fetch -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/fetch.err )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/write -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/write.err )
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $R/gpurun_out/prof/mfma -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/mfma.err )
( cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/prof/lds -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/lds.err )
( cd /tmp && rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/prof/wait -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/wait.err )
python tools/phase_profile.py 1024 > gpurun_out/${TAG}_phase_cycles_cold.json 2> gpurun_out/phase.err
python tools/phase_profile.py 1024 mpc > gpurun_out/${TAG}_phase_cycles_mpc.json 2>> gpurun_out/phase.err
( cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/prof/occ -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/occ.err )
python bench.py --agents 4096 --no-cpu --no-extras > gpurun_out/${TAG}_bench_n1_4096agents.json 2> gpurun_out/bench_4096.err
python bench.py --agents 256 --no-cpu --no-extras > gpurun_out/${TAG}_bench_n1_256agents.json 2> gpurun_out/bench_256.err
python bench.py --tol 1e-6 --no-cpu --no-extras > gpurun_out/${TAG}_bench_n1_tol1e-6.json 2> gpurun_out/bench_tol.err
python bench.py --workload quadrotor --agents 4096 --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_quadrotor_4096.json 2> gpurun_out/bench_quadrotor4096.err
python bench.py --workload holonomic3d --agents 8192 --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_holonomic3d_8192.json 2> gpurun_out/bench_h3d8192.err
python bench.py --workload formation --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_formation_n1.json 2> gpurun_out/bench_formation.err
python bench.py --workload rendezvous --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_rendezvous.json 2> gpurun_out/bench_rendezvous.err
python bench.py --workload quadrotor --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_quadrotor.json 2> gpurun_out/bench_quadrotor.err
python bench.py --workload holonomic3d --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_holonomic3d.json 2> gpurun_out/bench_holonomic3d.err
python tools/cpu_pool_sweep.py > gpurun_out/${TAG}_cpu_pool_sweep.txt 2>&1
python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/bench_final.err
ls gpurun_out/prof/*/ | head -30
