#!/bin/bash
mkdir -p gpurun_out/r05l
python tests/gpu_debug/uniform_sign_timing.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 > gpurun_out/r05l/uniform_levels.txt
cat gpurun_out/r05l/uniform_levels.txt
bash profiles/collect.sh r05
