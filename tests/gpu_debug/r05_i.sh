#!/bin/bash
mkdir -p gpurun_out/r05i
timeout 3000 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r05i/pytest_gpu.txt 2>&1
tail -16 gpurun_out/r05i/pytest_gpu.txt
