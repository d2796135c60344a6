"""Timeline of the LAST host-pointer call in a rocprofv3 --kernel-trace --memory-copy-trace directory: kernels and copies in start order, relative to the call's first event
(events after the last gap of more than 20 ms).   python3 tests/gpu_debug/trace_timeline.py <dir>"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0].replace("plume::", ""), ""))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "copy")), r.get("Bytes", r.get("Size", ""))))
ev.sort()
cut = 0
for i in range(1, len(ev)):
    if ev[i][0] - max(e[1] for e in ev[max(0, i - 50):i]) > 20_000_000:
        cut = i
ev = ev[cut:]
t0 = ev[0][0]
busy_k = 0
last_end = t0
for s, e, kind, name, extra in ev:
    print(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e3:9.1f} us  {kind} {name[:60]} {extra}")
print("span ms", (max(e[1] for e in ev) - t0) / 1e6, "events", len(ev))
# union of kernel intervals
iv = sorted((s, e) for s, e, k, *_ in ev if k == "K")
tot = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        tot += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
tot += ce - cs
print("kernel-busy ms (union)", tot / 1e6)
