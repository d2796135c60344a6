#!/bin/bash
# Round 6, last pass: the three unprofiled bench lines again (bench.py's notes changed after the collection; the counter summary of the same build is in profiles/),
# the signer's host call on one and two lanes on the same box, and the soaks with fresh seeds.
cd "$GRAFT_REPO_ROOT" || exit 1
R=r06
export TMPDIR=/tmp
mkdir -p gpurun_out/r06u
python3 bench.py > gpurun_out/bench_$R.log 2> gpurun_out/bench_$R.err
python3 bench.py --config 3 > gpurun_out/bench_${R}_sign.log 2> gpurun_out/bench_${R}_sign.err
python3 bench.py --config 4 --gpus 1 --log2-batch 19 --no-cpu-baseline > gpurun_out/bench_${R}_c4share.log 2> gpurun_out/bench_${R}_c4share.err
tail -1 gpurun_out/bench_$R.log | cut -c1-300
{ for l in 1 2 1 2; do PLUME_HOST_SIGN_LANES=$l timeout 600 python3 tests/gpu_debug/r06_sign_lanes.py 19 20 2>&1 | grep "sign lanes"; done; } | tee gpurun_out/r06u/sign_lanes_same_box.txt
{ echo "== python3 tests/gpu_debug/soak.py 6 18 (seeds from 100)"; ( time timeout 900 python3 tests/gpu_debug/soak.py 6 18 100 ) 2>&1 | grep -v amdgpu.ids
  echo "== python3 tests/gpu_debug/soak_ragged.py 500 11"; ( time timeout 1200 python3 tests/gpu_debug/soak_ragged.py 500 11 ) 2>&1 | grep -v amdgpu.ids | tail -n 12; } | tee gpurun_out/r06u/soak.txt | tail -n 12
