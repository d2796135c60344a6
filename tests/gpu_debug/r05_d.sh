#!/bin/bash
mkdir -p gpurun_out/r05d
for q in 0 8 16; do
  if [ $q != 0 ]; then export GPU_MAX_HW_QUEUES=$q; fi
  echo "##### GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-default}" >> gpurun_out/r05d/walls.txt
  python tests/gpu_debug/host_trace.py 20 both 2>&1 | grep -E "wall|device-resident" | cut -c1-60 >> gpurun_out/r05d/walls.txt
  echo "## sign two lanes" >> gpurun_out/r05d/walls.txt
  PLUME_HOST_SIGN_LANES=2 python tests/gpu_debug/host_trace.py 20 sign 2>&1 | grep -E "wall" >> gpurun_out/r05d/walls.txt
  PLUME_HOST_TRACE=1 PLUME_HOST_SIGN_LANES=2 python tests/gpu_debug/host_trace.py 20 both > gpurun_out/r05d/trace_q$q.txt 2>&1
done
cat gpurun_out/r05d/walls.txt
