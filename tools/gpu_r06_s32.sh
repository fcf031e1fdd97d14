#!/bin/bash
# a step as prediction + shift + solve in ONE launch (omgx_batch_rollout with one step) against the separate launches of BatchP2P.step
python tools/step_vs_rollout1.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s32.txt
