#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06d
export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in ship b8t; do
    if [ $v = ship ]; then unset PLUME_HIP_LIB; else export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_$v.so; fi
    echo -n "$v rep$rep: "; timeout 300 python3 tests/gpu_debug/r06_exp_b8.py 2>&1 | tail -n 1
  done
done | tee gpurun_out/r06d/exp_b8t.txt
unset PLUME_HIP_LIB
( time timeout -s ABRT 600 python3 -X faulthandler bench.py > gpurun_out/r06d/bench_default.json 2> gpurun_out/r06d/bench_default.err ) 2>&1 | tail -n 4; echo "bench default rc=$?"
( time timeout -s ABRT 800 python3 -X faulthandler bench.py --steps 10 --warmup 3 > gpurun_out/r06d/bench_10.json 2> gpurun_out/r06d/bench_10.err ) 2>&1 | tail -n 4
tail -n 30 gpurun_out/r06d/bench_default.err gpurun_out/r06d/bench_10.err | cut -c1-250
