#!/bin/bash
# phase-cycle profiles of the spill classes (cold batch and receding-horizon steps)
mkdir -p gpurun_out/s18
for sc in quadrotor_p2p holonomic3d_p2p; do
  OMGX_SCENARIO=$sc timeout 600 python tools/phase_profile.py 1024 > gpurun_out/s18/phase_cold_$sc.json 2> gpurun_out/s18/phase_cold_$sc.err
  OMGX_SCENARIO=$sc timeout 600 python tools/phase_profile.py 1024 mpc > gpurun_out/s18/phase_mpc_$sc.json 2> gpurun_out/s18/phase_mpc_$sc.err
done
tail -3 gpurun_out/s18/*.err
