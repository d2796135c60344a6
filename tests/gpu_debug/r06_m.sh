#!/bin/bash
# round 6: the runtime's hardware-queue pool (GPU_MAX_HW_QUEUES, 4 by default) against batches in flight -- the headline run and the small-call sweep
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06m
export TMPDIR=/tmp
for rep in 1 2 3; do
  for q in default 8 16; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues=$q rep$rep', d['value'], d['ms_per_step'], 'serial', d['ms_per_step_serial'], d['in_flight']['stage_ms_in_flight'])"
  done
done | tee gpurun_out/r06m/queues_headline.txt
for q in default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  echo "== GPU_MAX_HW_QUEUES=$q"; timeout 600 python3 tests/gpu_debug/r06_small_sweep.py b 2>&1 | grep "2\^"
done | tee gpurun_out/r06m/queues_small.txt
for q in default 8; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 300 python3 bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('sign queues=$q', d['value'], d['ms_per_step'], 'serial', d['ms_per_step_serial'])"
done | tee -a gpurun_out/r06m/queues_headline.txt
