#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06f
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06f/prof -o b8 -- python3 $GRAFT_REPO_ROOT/bench.py --in-flight 1 --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-probe > $GRAFT_REPO_ROOT/gpurun_out/r06f/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
head -n 14 gpurun_out/r06f/prof/b8_kernel_stats.csv | cut -c1-150
for v in new d2 d1; do
  if [ $v = new ]; then unset PLUME_HIP_LIB; else export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_$v.so; fi
  echo -n "$v: "; timeout 300 python3 tests/gpu_debug/r06_exp_b8.py 2>&1 | tail -n 1
done | tee gpurun_out/r06f/passd_waves.txt
