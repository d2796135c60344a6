#!/bin/bash
# round 6: soaks with the nonce-zero plants in every fuzz batch; and the first GPU call's sequence again (twice), whose bench run once never came back
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06k
export TMPDIR=/tmp
( time timeout 1500 python3 tests/gpu_debug/soak.py 6 18 ) > gpurun_out/r06k/soak.txt 2>&1; echo "soak rc=$?"; tail -n 3 gpurun_out/r06k/soak.txt
( time timeout 1200 python3 tests/gpu_debug/soak_ragged.py 400 5 ) > gpurun_out/r06k/soak_ragged.txt 2>&1; echo "ragged rc=$?"; tail -n 3 gpurun_out/r06k/soak_ragged.txt
for i in 1 2; do
  timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > gpurun_out/r06k/seq_pytest_$i.log 2>&1
  timeout 600 python3 tests/gpu_debug/r06_small_sweep.py ab > gpurun_out/r06k/seq_sweep_$i.txt 2>&1
  ( time timeout -s ABRT 400 python3 -X faulthandler bench.py --steps 10 --warmup 3 > gpurun_out/r06k/seq_bench_$i.json 2> gpurun_out/r06k/seq_bench_$i.err ) 2>&1 | grep real; echo "seq $i bench rc=$?"
done
