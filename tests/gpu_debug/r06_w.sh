#!/bin/bash
# Round 6: long soaks of the final build (fresh seeds again)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r06w
{ echo "== python3 tests/gpu_debug/soak.py 50 18 200"; ( time timeout 1500 python3 tests/gpu_debug/soak.py 50 18 200 ) 2>&1 | grep -v amdgpu.ids | tail -n 8
  echo "== python3 tests/gpu_debug/soak.py 3 20 300   (2^20-item batches)"; ( time timeout 900 python3 tests/gpu_debug/soak.py 3 20 300 ) 2>&1 | grep -v amdgpu.ids | tail -n 8
  echo "== python3 tests/gpu_debug/soak_ragged.py 2000 12"; ( time timeout 1500 python3 tests/gpu_debug/soak_ragged.py 2000 12 ) 2>&1 | grep -v amdgpu.ids | tail -n 6; } | tee gpurun_out/r06w/soak_long.txt | tail -n 30
