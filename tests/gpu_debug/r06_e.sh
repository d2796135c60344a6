#!/bin/bash
# round 6: base-8 digits for equation 2 -- the whole GPU suite, then an A/B against the build of the commit before (libplume_hip_prev.so), alternating on this box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06e
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r06e/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/r06e/rc.txt
tail -n 15 gpurun_out/r06e/pytest_gpu.log
for rep in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_prev.so; else unset PLUME_HIP_LIB; fi
    timeout 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v rep$rep', d['value'], d['ms_per_step'], 'serial', d['ms_per_step_serial'], d['stage_ms'], 'cyc/item', r.get('cycles_per_item'), 'GHz', r.get('clock_ghz_in_kernel_this_run'))"
  done
done | tee gpurun_out/r06e/ab_base8.txt
unset PLUME_HIP_LIB
timeout 600 python3 bench.py > gpurun_out/r06e/bench_default.json 2> gpurun_out/r06e/bench_default.err; echo "bench rc=$?" | tee -a gpurun_out/r06e/rc.txt
for v in prev new; do
  if [ $v = prev ]; then export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_prev.so; else unset PLUME_HIP_LIB; fi
  timeout 300 python3 bench.py --config 4 --log2-batch 19 --steps 6 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v V2 2^19', d['value'], d['ms_per_step'], d['stage_ms'])"
  timeout 300 python3 bench.py --config 2 --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v 2^16', d['value'], d['ms_per_step'], d['stage_ms'])"
done | tee -a gpurun_out/r06e/ab_base8.txt
