#!/bin/bash
mkdir -p gpurun_out/r05n
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r05n/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r05n/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05n/smoke.txt 2>&1; tail -1 gpurun_out/r05n/smoke.txt
bash profiles/collect.sh r05
