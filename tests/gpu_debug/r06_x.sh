#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r06x
{ echo "== python3 tests/gpu_debug/r06_sign_soak.py 40 3"; ( time timeout 1500 python3 tests/gpu_debug/r06_sign_soak.py 40 3 ) 2>&1 | grep -v amdgpu.ids | tail -n 10
  echo "== python3 tests/gpu_debug/soak_ragged.py 1500 21"; ( time timeout 1500 python3 tests/gpu_debug/soak_ragged.py 1500 21 ) 2>&1 | grep -v amdgpu.ids | tail -n 6; } | tee gpurun_out/r06x/soak_sign.txt | tail -n 24
