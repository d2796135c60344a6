#!/bin/bash
# round 5, build with the checked half-GCD pair: GPU suite, smoke, then the collection for the new build id and a soak
mkdir -p gpurun_out/r05w
python -m pytest tests -m gpu -q -x --durations=3 > gpurun_out/r05w/pytest_gpu.txt 2>&1; grep -E "passed|failed" gpurun_out/r05w/pytest_gpu.txt | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05w/smoke.txt 2>&1; tail -1 gpurun_out/r05w/smoke.txt
bash profiles/collect.sh r05
timeout 900 python3 tests/gpu_debug/soak.py 10 19 > gpurun_out/r05w/soak.txt 2>&1; tail -2 gpurun_out/r05w/soak.txt
