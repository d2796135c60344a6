#!/bin/bash
# round 6: the host-pointer pipeline's piece schedules again, with the hardware-queue pool at 8 (the Python package sets it) and at the default 4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06q
export TMPDIR=/tmp
for q in 8 4; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q timeout 600 python3 tests/gpu_debug/host_sched_r05.py verify 2>&1 | grep -E "verify|lanes"
  GPU_MAX_HW_QUEUES=$q timeout 600 python3 tests/gpu_debug/host_sched_r05.py sign 2>&1 | grep -E "sign lanes"
done | tee gpurun_out/r06q/host_sched.txt
