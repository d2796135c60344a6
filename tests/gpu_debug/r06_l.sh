#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06l
export TMPDIR=/tmp PLUME_IN_FLIGHT_MIN=0
for m in lanes engines; do python3 tests/gpu_debug/r06_small_overlap.py 12 $m 2>&1 | grep "ms per call"; done | tee gpurun_out/r06l/overlap.txt
GPU_MAX_HW_QUEUES=8 python3 tests/gpu_debug/r06_small_overlap.py 12 lanes 2>&1 | grep "ms per call" | sed 's/^/GPU_MAX_HW_QUEUES=8 /' | tee -a gpurun_out/r06l/overlap.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06l/trace -o t -- python3 $GRAFT_REPO_ROOT/tests/gpu_debug/r06_small_overlap.py 12 lanes > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/r06l/trace/t_kernel_trace.csv")))
rows = [r for r in rows if "plume::k_verify" in r["Kernel_Name"] or "plume::k_tab" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "kernels; queue ids:", collections.Counter(r["Queue_Id"] for r in rows))
tail = rows[-64:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail[:40]:
    print(r["Queue_Id"], r["Kernel_Name"].split("(")[0][7:30].ljust(24), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3)
PY
