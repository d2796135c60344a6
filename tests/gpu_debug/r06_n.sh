#!/bin/bash
# round 6: the lanes of plume_set_in_flight on their own streams -- the tests that use lanes, the small-call sweep, the headline and the signer against the build before
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06n
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_round3.py -x -q -m gpu -k "lanes or stage_timing or contexts or nonce or context_says or fixed_tables" > gpurun_out/r06n/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 4 gpurun_out/r06n/pytest.log
GPU_MAX_HW_QUEUES=8 timeout 600 python3 tests/gpu_debug/r06_small_sweep.py b 2>&1 | grep "2\^" | tee gpurun_out/r06n/sweep_b.txt
for rep in 1 2 3; do
  for v in prev new; do
    if [ $v = new ]; then unset PLUME_HIP_LIB; else export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_$v.so; fi
    timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v verify rep$rep', d['value'], d['ms_per_step'], 'serial', d['ms_per_step_serial'])"
    timeout 300 python3 bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v sign rep$rep', d['value'], d['ms_per_step'], 'serial', d['ms_per_step_serial'])"
  done
done | tee gpurun_out/r06n/ab.txt
