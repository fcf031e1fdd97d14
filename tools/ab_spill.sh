python bench.py --workload holonomic3d --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-200
python bench.py --workload quadrotor --steps 5 --warmup 2 2>&1 | tail -1 | cut -c1-200
OMGX_SCENARIO=holonomic3d_p2p python tools/phase_profile.py 256 > gpurun_out/phase_h3d.json 2> gpurun_out/phase_h3d.err
OMGX_SCENARIO=quadrotor_p2p python tools/phase_profile.py 256 > gpurun_out/phase_quad.json 2> gpurun_out/phase_quad.err
python -m pytest tests/test_gpu_spill.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
