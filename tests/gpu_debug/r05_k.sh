#!/bin/bash
mkdir -p gpurun_out/r05k
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r05k/pytest_gpu.txt 2>&1
tail -8 gpurun_out/r05k/pytest_gpu.txt
