#!/bin/bash
# Round 6: does the larger hardware-queue pool cost the 2^20 line anything?  bench.py with GPU_MAX_HW_QUEUES=4 (the runtime's default) and 8, alternating on one box.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r06v
{ for rep in 1 2 3 4; do for q in 4 8; do
    GPU_MAX_HW_QUEUES=$q timeout 300 python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extras --no-probe 2>/dev/null | tail -n 1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('queues $q  in flight %.2f M/s (%.3f ms)  serial %.2f M/s  msm kernel %.3f ms  clock %.3f GHz  cycles/item %.2f' % (d['value'] / 1e6, d['ms_per_step'], d['value_serial'] / 1e6, r['kernel_ms'], r['clock_ghz_in_kernel_this_run'], r['cycles_per_item']))"
  done; done; } | tee gpurun_out/r06v/hw_queues_2p20.txt
