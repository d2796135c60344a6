#!/bin/bash
# round 6: pricing base-8 digits for equation 2 with two timing experiments, alternating with the shipped build on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06b
for rep in 1 2 3; do
  for v in ship b8m b8t; do
    if [ $v = ship ]; then unset PLUME_HIP_LIB; else export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_$v.so; fi
    echo -n "$v rep$rep: "; timeout 600 python3 tests/gpu_debug/r06_exp_b8.py 2>&1 | tail -1
  done
done | tee gpurun_out/r06b/exp_b8.txt
