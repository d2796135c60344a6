#!/bin/bash
mkdir -p gpurun_out/r05c
PLUME_HOST_TRACE=1 python tests/gpu_debug/host_trace.py 20 both > gpurun_out/r05c/trace_default.txt 2>&1
PLUME_HOST_TRACE=1 PLUME_HOST_SIGN_LANES=2 python tests/gpu_debug/host_trace.py 20 sign > gpurun_out/r05c/trace_sign_two_lanes.txt 2>&1
timeout 2400 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/r05c/pytest_r5.txt 2>&1
tail -5 gpurun_out/r05c/pytest_r5.txt
