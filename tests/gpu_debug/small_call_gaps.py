"""From a rocprofv3 --kernel-trace directory of tests/gpu_debug/small_call_trace.py: the kernels of the LAST three calls in start order with the idle gap in front of each,
and per kernel name the median duration and the median gap in front of it.    python3 tests/gpu_debug/small_call_gaps.py <dir>"""
import csv, glob, statistics, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("plume::", "")))
ev.sort()
fin = [i for i, e in enumerate(ev) if e[2] == "k_verify_finalize"]
lo = fin[-4] + 1
t0 = ev[lo][0]
for i in range(lo, fin[-1] + 1):
    s, e, name = ev[i]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  gap in front {(s - ev[i - 1][1]) / 1e3:6.1f} us  {name}")
dur, gap = {}, {}
for i in range(fin[3] + 1, fin[-1] + 1):           # (the first calls warm up)
    s, e, name = ev[i]
    dur.setdefault(name, []).append((e - s) / 1e3); gap.setdefault(name, []).append((s - ev[i - 1][1]) / 1e3)
print("\nmedians over the calls after the third:")
for name in dur:
    print(f"  {name:28s} {statistics.median(dur[name]):8.1f} us   gap in front {statistics.median(gap[name]):6.1f} us   ({len(dur[name])} launches)")
per_call = [(ev[fin[k]][1] - ev[fin[k - 1]][1]) / 1e3 for k in range(4, len(fin))]
print(f"  call period (finalize end to finalize end): median {statistics.median(per_call):.1f} us")
