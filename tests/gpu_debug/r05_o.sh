#!/bin/bash
# round 5, after the host-side fixes found by tests/hostsim (lane turn of a refused call, per-device lock of the generator tables): the GPU suite and the smoke run
mkdir -p gpurun_out/r05o
python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r05o/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r05o/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05o/smoke.txt 2>&1; tail -1 gpurun_out/r05o/smoke.txt
python bench.py > gpurun_out/r05o/bench.txt 2> gpurun_out/r05o/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05o/bench.txt').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source']['same_build'])
PY
