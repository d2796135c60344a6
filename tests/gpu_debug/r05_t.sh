#!/bin/bash
# round 5, final tree: GPU suite, smoke, the default bench line
mkdir -p gpurun_out/r05t
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r05t/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r05t/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05t/smoke.txt 2>&1; tail -1 gpurun_out/r05t/smoke.txt
( time python bench.py > gpurun_out/r05t/bench.txt 2> gpurun_out/r05t/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05t/bench.txt').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['value_serial'], d['roofline']['frac'], d['roofline']['traffic_source']['same_build'], d['other_workloads']['verify_v1_2p16']['ms_per_batch'], d['e2e_host_pinned']['verify_v1']['frac_of_value_serial'], d['e2e_host_pinned']['sign_v1']['frac_of_device_resident_sign'])
PY
