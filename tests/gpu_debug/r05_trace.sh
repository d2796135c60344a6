#!/bin/bash
# round 5: kernel traces of back-to-back small verify calls (library defaults: no stage events), 2^16 and 2^14 items, and with the stage events on for the comparison
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for cfg in "16 0" "14 0" "16 1"; do
  set -- $cfg
  rm -rf gpurun_out/sc_$1_$2
  PLUME_STAGE_TIMES=$2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sc_$1_$2 -o sc -- python3 tests/gpu_debug/small_call_trace.py $1 > gpurun_out/sc_$1_$2.log 2>&1
  echo "=== 2^$1 items per call, stage-timing events $( [ $2 = 1 ] && echo on || echo 'off (the default)')"
  python3 tests/gpu_debug/small_call_gaps.py gpurun_out/sc_$1_$2
done > gpurun_out/small_call_gaps.txt 2>&1
tail -45 gpurun_out/small_call_gaps.txt
