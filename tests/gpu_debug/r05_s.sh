#!/bin/bash
# round 5: same-box A/B of the round-5 build (in-tree, A) against the round-4 kernels and C ABI (commit d5082ac's csrc/, built with the same hipcc as libplume_hip_r04.so):
# alternating runs of the metric workload, in flight and serial, and of the signer; then a longer soak
mkdir -p gpurun_out/r05s
{
for rep in 1 2 3; do
  for v in A r04; do
    if [ $v = A ]; then unset PLUME_HIP_LIB; else export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_$v.so; fi
    python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v rep$rep verify', d['value'], d['ms_per_step'], 'serial', d.get('value_serial'), d.get('ms_per_step_serial'), d['stage_ms'], 'frac', d['roofline']['frac'])"
    python bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v rep$rep sign  ', d['value'], d['ms_per_step'], 'serial', d.get('value_serial'), d.get('ms_per_step_serial'), d['stage_ms'])"
  done
done
} > gpurun_out/r05s/ab_vs_r04.txt 2>&1
unset PLUME_HIP_LIB
cat gpurun_out/r05s/ab_vs_r04.txt | cut -c1-260
timeout 1200 python3 tests/gpu_debug/soak.py 24 19 > gpurun_out/r05s/soak.txt 2>&1; tail -3 gpurun_out/r05s/soak.txt
