#!/bin/bash
mkdir -p gpurun_out/r05j
python tests/gpu_debug/eq1_ab.py 20 2>&1 | head -4 > gpurun_out/r05j/eq1_ab_2p20.txt
python tests/gpu_debug/eq1_ab.py 16 2>&1 | head -2 > gpurun_out/r05j/eq1_ab_2p16.txt
cat gpurun_out/r05j/eq1_ab_2p20.txt gpurun_out/r05j/eq1_ab_2p16.txt
python bench.py --config 3 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('sign', d['value'], d['value_serial'], d['stage_ms'])" > gpurun_out/r05j/sign.txt 2>&1
cat gpurun_out/r05j/sign.txt
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r05j/pytest_gpu.txt 2>&1
tail -6 gpurun_out/r05j/pytest_gpu.txt
