#!/bin/bash
mkdir -p gpurun_out/r05y
python3 tests/gpu_debug/pair_ab.py > gpurun_out/r05y/pair_ab.txt 2>&1; tail -14 gpurun_out/r05y/pair_ab.txt
python -m pytest tests -m gpu -q -x --durations=3 > gpurun_out/r05y/pytest_gpu.txt 2>&1; grep -E "passed|failed" gpurun_out/r05y/pytest_gpu.txt | tail -1
timeout 600 python3 tests/gpu_debug/soak_ragged.py 600 77 > gpurun_out/r05y/ragged.txt 2>&1; tail -1 gpurun_out/r05y/ragged.txt
