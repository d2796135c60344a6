#!/bin/bash
mkdir -p gpurun_out/r05f
timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r05f/pytest_gpu.txt 2>&1
tail -25 gpurun_out/r05f/pytest_gpu.txt
