#!/bin/bash
mkdir -p gpurun_out/r05h
python tests/gpu_debug/eq1_ab.py 20 2>&1 | head -4 > gpurun_out/r05h/eq1_ab_2p20.txt
cat gpurun_out/r05h/eq1_ab_2p20.txt
