#!/bin/bash
# round 6: the signer's chains cut at 32 bits (four tables per item) -- sign tests, A/B against the build before; comb / generator-window widths against the kernel's clock
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06i
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -x -q -m gpu -k "sign or uniform or arkworks or config3 or der or every_item" > gpurun_out/r06i/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/r06i/pytest.log
for rep in 1 2 3; do
  for v in prev new; do
    if [ $v = new ]; then unset PLUME_HIP_LIB; else export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_$v.so; fi
    timeout 300 python3 bench.py --config 3 --in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-probe 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v sign rep$rep', d['value'], d['ms_per_step'], d['stage_ms'])"
  done
done | tee gpurun_out/r06i/ab_sign.txt
for rep in 1 2; do
  for v in new comb16 comb14; do
    if [ $v = new ]; then unset PLUME_HIP_LIB; else export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_$v.so; fi
    echo -n "$v: "; timeout 300 python3 tests/gpu_debug/r06_exp_b8.py 2>&1 | tail -n 1
  done
  for v in new gw20 gw16; do
    if [ $v = new ]; then unset PLUME_HIP_LIB; else export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_$v.so; fi
    timeout 300 python3 bench.py --config 4 --gpus 1 --log2-batch 19 --in-flight 1 --steps 8 --warmup 2 --no-cpu-baseline --no-extras --no-probe 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v V2 2^19 rep$rep', d['value'], d['ms_per_step'], d['stage_ms'], d['roofline'].get('clock_ghz_in_kernel_this_run'))"
  done
done | tee gpurun_out/r06i/widths.txt
