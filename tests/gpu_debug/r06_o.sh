#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06o
export TMPDIR=/tmp
python3 tests/gpu_debug/r06_small_overlap.py 12 lanes 2>&1 | grep "ms per call" | tee gpurun_out/r06o/overlap.txt
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06o/trace -o t -- python3 $GRAFT_REPO_ROOT/tests/gpu_debug/r06_small_overlap.py 12 lanes > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/r06o/trace/t_kernel_trace.csv")))
rows = [r for r in rows if "plume::k_verify" in r["Kernel_Name"] or "plume::k_tab" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(len(rows), "kernels; queue ids:", collections.Counter(r["Queue_Id"] for r in rows))
tail = rows[-48:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail[:32]:
    print(r["Queue_Id"], r["Kernel_Name"].split("(")[0][7:30].ljust(24), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3)
PY
