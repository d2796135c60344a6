#!/bin/bash
# round 5, build with the fused scalar stage: GPU suite, smoke, small-call A/B again (now with the same-stream workspace rule), then the collection for the new build id
mkdir -p gpurun_out/r05q
python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r05q/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r05q/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05q/smoke.txt 2>&1; tail -1 gpurun_out/r05q/smoke.txt
python3 tests/gpu_debug/small_call_ab.py > gpurun_out/r05q/small_call_ab.txt 2>&1; grep "2^16" gpurun_out/r05q/small_call_ab.txt
bash profiles/collect.sh r05
