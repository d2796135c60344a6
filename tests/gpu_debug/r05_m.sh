#!/bin/bash
mkdir -p gpurun_out/r05m
python tests/gpu_debug/uniform_sign_timing.py 2>&1 | head -3 | cut -c1-330 > gpurun_out/r05m/uniform_levels.txt
cat gpurun_out/r05m/uniform_levels.txt
timeout 1200 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "uniform" > gpurun_out/r05m/pytest_uniform.txt 2>&1; tail -3 gpurun_out/r05m/pytest_uniform.txt
