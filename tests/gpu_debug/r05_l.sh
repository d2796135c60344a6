#!/bin/bash
mkdir -p gpurun_out/r05l
python tests/gpu_debug/uniform_sign_timing.py > gpurun_out/r05l/uniform_levels.txt 2>&1
cat gpurun_out/r05l/uniform_levels.txt | cut -c1-330
bash profiles/collect.sh r05
