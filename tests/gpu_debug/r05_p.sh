#!/bin/bash
# round 5: small-call A/B (stage events, scalar stage fused into the two-role ingest kernel), then the GPU suite on the new build
mkdir -p gpurun_out/r05p
python3 tests/gpu_debug/small_call_ab.py > gpurun_out/r05p/small_call_ab.txt 2>&1; tail -16 gpurun_out/r05p/small_call_ab.txt
python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r05p/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r05p/pytest_gpu.txt
