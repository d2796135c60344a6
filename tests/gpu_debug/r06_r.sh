#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06r
export TMPDIR=/tmp
{ timeout 600 python3 tests/gpu_debug/r06_host_sched.py verify 2>&1 | grep "lanes"; timeout 600 python3 tests/gpu_debug/r06_host_sched.py sign 2>&1 | grep "lanes";
  timeout 300 python3 bench.py --in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-probe 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('device-resident verify', d['ms_per_step'])"
  timeout 300 python3 bench.py --config 3 --in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-probe 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('device-resident sign', d['ms_per_step'])"; } | tee gpurun_out/r06r/host_sched2.txt
