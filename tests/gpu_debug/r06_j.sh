#!/bin/bash
# round 6: the whole GPU suite on the tree as it stands, then new vs prev (fb6c178) for the signer (generic K = 2 chains) and the verifier
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06j
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r06j/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/r06j/rc.txt
tail -n 6 gpurun_out/r06j/pytest_gpu.log
for rep in 1 2 3; do
  for v in prev new; do
    if [ $v = new ]; then unset PLUME_HIP_LIB; else export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_$v.so; fi
    timeout 300 python3 bench.py --config 3 --in-flight 1 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-probe 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v sign rep$rep', d['value'], d['ms_per_step'], d['stage_ms'])"
    timeout 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-probe 2>/dev/null | tail -n 1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v verify rep$rep', d['value'], d['ms_per_step'], d['ms_per_step_serial'], d['stage_ms'])"
  done
done | tee gpurun_out/r06j/ab.txt
unset PLUME_HIP_LIB
for i in 1 2 3; do ( time timeout 600 python3 bench.py > gpurun_out/r06j/bench_default_$i.json 2> gpurun_out/r06j/bench_default_$i.err ) 2>&1 | grep real; done
