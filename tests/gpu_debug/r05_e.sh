#!/bin/bash
mkdir -p gpurun_out/r05e
python tests/gpu_debug/host_sched_r05.py sign > gpurun_out/r05e/sign.txt 2>&1
python tests/gpu_debug/host_sched_r05.py verify > gpurun_out/r05e/verify.txt 2>&1
PLUME_HOST_TRACE=1 PLUME_HOST_SIGN_LANES=2 python tests/gpu_debug/host_trace.py 20 both > gpurun_out/r05e/trace.txt 2>&1
cat gpurun_out/r05e/sign.txt gpurun_out/r05e/verify.txt
