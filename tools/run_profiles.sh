# One collection run of the evidence under profiles/ (round tag = $1, default r03).  Run on the GPU box:
#   gpurun -- 'bash tools/run_profiles.sh r03'      then   python tools/profile_summary.py r03
# The stats pass and every PMC pass are separate rocprofv3 runs (counters are never combined with trace domains).
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
R=$GRAFT_REPO_ROOT
# the stats pass profiles the DEFAULT bench (two half-launches per step); the counter passes run one launch per step (--streams 1):
# launches 0-4 of the process are then cold solves of the 1024-agent batch, the rest receding-horizon steps of all 1024 agents
B1="python $R/bench.py --no-cpu --no-extras --no-parity"
B="python $R/bench.py --streams 1 --no-cpu --no-extras --no-parity"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -- $B1 > $R/gpurun_out/prof/bench_stats.json 2> $R/gpurun_out/prof/stats.err )
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/fetch -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/fetch.err )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/write -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/write.err )
( cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $R/gpurun_out/prof/mfma -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/mfma.err )
( cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/prof/lds -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/lds.err )
( cd /tmp && rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/prof/wait -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/wait.err )
python tools/phase_profile.py 1024 > gpurun_out/${TAG}_phase_cycles_cold.json 2> gpurun_out/phase.err
python tools/phase_profile.py 1024 mpc > gpurun_out/${TAG}_phase_cycles_mpc.json 2>> gpurun_out/phase.err
( cd /tmp && rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_CYCLES_SALU SQ_IFETCH --output-format csv -d $R/gpurun_out/prof/issue1 -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/issue1.err )
( cd /tmp && rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/prof/issue2 -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/issue2.err )
( cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/prof/occ -- $B --steps 5 --warmup 1 > /dev/null 2> $R/gpurun_out/prof/occ.err )
python bench.py --streams 1 --no-cpu --no-extras > gpurun_out/${TAG}_bench_n1_one_launch_per_step.json 2> gpurun_out/bench_1s.err
python bench.py --scaling strong --no-cpu --no-extras > gpurun_out/${TAG}_bench_n1_strong.json 2> gpurun_out/bench_strong.err
python bench.py --agents 4096 --no-cpu --no-extras > gpurun_out/${TAG}_bench_n1_4096agents.json 2> gpurun_out/bench_4096.err
python bench.py --agents 256 --no-cpu --no-extras > gpurun_out/${TAG}_bench_n1_256agents.json 2> gpurun_out/bench_256.err
python bench.py --tol 1e-6 --no-cpu --no-extras > gpurun_out/${TAG}_bench_n1_tol1e-6.json 2> gpurun_out/bench_tol.err
python bench.py --ipopt-defaults --no-cpu --no-extras > gpurun_out/${TAG}_bench_n1_ipopt_default_tolerances.json 2> gpurun_out/bench_ipd.err
python bench.py --workload quadrotor --agents 4096 --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_quadrotor_4096.json 2> gpurun_out/bench_quadrotor4096.err
python bench.py --workload holonomic3d --agents 8192 --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_holonomic3d_8192.json 2> gpurun_out/bench_h3d8192.err
python bench.py --workload formation --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_formation_n1.json 2> gpurun_out/bench_formation.err
python bench.py --workload rendezvous --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_rendezvous.json 2> gpurun_out/bench_rendezvous.err
python bench.py --workload quadrotor --steps 5 --warmup 2 > gpurun_out/${TAG}_bench_quadrotor.json 2> gpurun_out/bench_quadrotor.err
python bench.py --workload holonomic3d --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_holonomic3d.json 2> gpurun_out/bench_holonomic3d.err
python tools/cpu_pool_sweep.py > gpurun_out/${TAG}_cpu_pool_sweep.txt 2>&1
python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/bench_final.err
cp gpurun_out/prof/bench_stats.json gpurun_out/${TAG}_bench_n1_under_rocprof.json
python tools/profile_summary.py ${TAG} > gpurun_out/${TAG}_profile_summary.txt 2>&1; mkdir -p gpurun_out/profiles; python tools/profile_summary.py ${TAG} issue >> gpurun_out/${TAG}_profile_summary.txt 2>&1
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_pmc_*.json gpurun_out/profiles/ 2>/dev/null
tail -30 gpurun_out/${TAG}_profile_summary.txt
ls gpurun_out/prof/*/ | head -30
