"""Round 6: hunt for the intermittent stall of bench.py's secondary sections (one run in ~25 sits in them until the watchdog fires).  Runs bench.py N times with a short
watchdog and progress markers; prints per run the wall time and, for a run the watchdog ended, where it sat and the Python stacks.  Usage: python r06_stall_hunt.py [runs=25]"""
import json, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 25
extra = sys.argv[2:]
env = dict(os.environ, PLUME_BENCH_SECONDARY_TIMEOUT="100", PLUME_BENCH_PROGRESS="1")
stalls = 0
for i in range(runs):
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-probe", *extra], capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired as e:
        print(f"run {i}: NO LINE in 400 s; stderr tail:\n{(e.stderr or b'')[-3000:]}", flush=True)
        stalls += 1
        continue
    dt = time.time() - t0
    try:
        line = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        print(f"run {i}: rc {r.returncode}, no JSON; stderr tail:\n{r.stderr[-2000:]}", flush=True)
        continue
    if "watchdog" in line:
        stalls += 1
        print(f"run {i}: STALLED ({dt:.0f} s) in {line.get('watchdog_stalled_in')}\n--- stderr tail\n{r.stderr[-6000:]}\n---", flush=True)
    else:
        print(f"run {i}: ok {dt:.1f} s  value {line['value'] / 1e6:.2f} M/s", flush=True)
print(f"{stalls} stalled runs of {runs}")
