#!/bin/bash
# round 5, first GPU call: how downloads travel (d2h_probe), then the round's new GPU tests
mkdir -p gpurun_out/r05a
./tests/gpu_debug/d2h_probe > gpurun_out/r05a/d2h_probe.txt 2>&1
AMD_LOG_LEVEL=4 ./tests/gpu_debug/d2h_probe log 2>&1 | grep -E "^===|HSA Copy|Blit|copy engine|shader|Sdma|SDMA|staging" | cut -c1-300 > gpurun_out/r05a/d2h_log.txt
timeout 2400 python -m pytest tests/test_gpu_round5.py -x -q > gpurun_out/r05a/pytest_r5.txt 2>&1
tail -5 gpurun_out/r05a/pytest_r5.txt
cat gpurun_out/r05a/d2h_probe.txt
