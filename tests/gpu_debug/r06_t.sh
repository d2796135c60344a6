#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06t
export TMPDIR=/tmp
K=1024
uni() { python3 -c "print(','.join([str($1*1024)]*($2//$1)))"; }
{ for lg in 21 22; do
    T=$((1<<(lg-10)))
    echo "== 2^$lg"
    PLUME_HOST_SIGN_LANES=1 timeout 600 python3 tests/gpu_debug/r06_sign_lanes.py $lg 2>&1 | grep "sign lanes"
    PLUME_HOST_SIGN_LANES=2 timeout 600 python3 tests/gpu_debug/r06_sign_lanes.py $lg 2>&1 | grep "sign lanes"
    for p in 128 256 512; do echo "two lanes, uniform ${p}k"; PLUME_HOST_SCHEDULE=$(uni $p $T) PLUME_HOST_SIGN_LANES=2 timeout 600 python3 tests/gpu_debug/r06_sign_lanes.py $lg 2>&1 | grep "sign lanes"; done
    echo "two lanes, 64k x8 then 128k"; PLUME_HOST_SCHEDULE=$(python3 -c "print(','.join(['65536']*8+['131072']*(($T-512)//128)))") PLUME_HOST_SIGN_LANES=2 timeout 600 python3 tests/gpu_debug/r06_sign_lanes.py $lg 2>&1 | grep "sign lanes"
    echo "two lanes, 128k.. then 64k x8 at the end"; PLUME_HOST_SCHEDULE=$(python3 -c "print(','.join(['131072']*(($T-512)//128)+['65536']*8))") PLUME_HOST_SIGN_LANES=2 timeout 600 python3 tests/gpu_debug/r06_sign_lanes.py $lg 2>&1 | grep "sign lanes"
    echo "one lane, uniform 64k"; PLUME_HOST_SCHEDULE=$(uni 64 $T) PLUME_HOST_SIGN_LANES=1 timeout 600 python3 tests/gpu_debug/r06_sign_lanes.py $lg 2>&1 | grep "sign lanes"
  done; } | tee gpurun_out/r06t/sign_large.txt
