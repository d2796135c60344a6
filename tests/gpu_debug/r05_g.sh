#!/bin/bash
mkdir -p gpurun_out/r05g
python tests/gpu_debug/eq1_ab.py 20 > gpurun_out/r05g/eq1_ab_2p20.txt 2>&1
python tests/gpu_debug/eq1_ab.py 16 > gpurun_out/r05g/eq1_ab_2p16.txt 2>&1
cat gpurun_out/r05g/eq1_ab_2p20.txt gpurun_out/r05g/eq1_ab_2p16.txt
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r05g/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r05g/pytest_gpu.txt
