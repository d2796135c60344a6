#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06g
export TMPDIR=/tmp
for rep in 1 2; do
for v in prev new d2 d4; do
  if [ $v = new ]; then unset PLUME_HIP_LIB; else export PLUME_HIP_LIB=$PWD/zk-nullifier-sig_amd/libplume_hip_$v.so; fi
  echo -n "$v: "; timeout 300 python3 tests/gpu_debug/r06_exp_b8.py 2>&1 | tail -n 1
done
done | tee gpurun_out/r06g/passd_waves.txt
unset PLUME_HIP_LIB
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06g/prof -o b8 -- python3 $GRAFT_REPO_ROOT/bench.py --in-flight 1 --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-probe > $GRAFT_REPO_ROOT/gpurun_out/r06g/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r06g/prof/b8_kernel_stats.csv")):
    if "tab" in r["Name"] or "verify" in r["Name"]:
        print(r["Name"][:40], r["Calls"], round(float(r["AverageNs"])/1e6,4), round(float(r["MinNs"])/1e6,4), round(float(r["MaxNs"])/1e6,4))
PY
