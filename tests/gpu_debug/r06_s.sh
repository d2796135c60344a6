#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06s
export TMPDIR=/tmp
{ for rep in 1 2; do for l in 1 2; do PLUME_HOST_SIGN_LANES=$l timeout 600 python3 tests/gpu_debug/r06_sign_lanes.py 17 18 19 20 22 2>&1 | grep "sign lanes"; done; done
  timeout 900 python3 -m pytest tests -m gpu -x -q -k "host or sign or pipeline or pinned" 2>&1 | tail -n 3; } | tee gpurun_out/r06s/sign_lanes.txt
