"""(round 5) A/B of the verifier's first equation: long form (PLUME_EQ1_SHORT=0) against the short form (plume_eis.h) on one box, one process per setting, alternating.
Device-resident 2^20 V1 verifies (1/16 corrupted), serial stage times (median of the timed calls) and the rate with two batches in flight."""
import os; os.environ.setdefault("PLUME_STAGE_TIMES", "1")   # the stage-timing events are off by default since library 0.5; this script reads them
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
CHILD = r'''
import sys, time, json
import numpy as np, torch
sys.path.insert(0, %r)
import zk_nullifier_sig_amd as plume
from tests import synth
n = 1 << int(sys.argv[1])
eng = plume.Engine(0)
b = synth.sign_inputs(n)
sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, sg)
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
d = {k: t(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
off = t(v["off"].view(np.int64)); mb = int(v["off"][-1])
ok = torch.zeros(n, dtype=torch.uint8, device=dev); ok2 = torch.zeros(n, dtype=torch.uint8, device=dev)
call = lambda o, st=None: eng.verify_batch_device(1, n, d["msgs"], off, mb, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], o, stream=st)
for _ in range(3): call(ok)
torch.cuda.synchronize()
acc = {}
R = 7
t0 = time.perf_counter()
for _ in range(R):
    call(ok); torch.cuda.synchronize()
    for k, ms in eng.last_stage_times(): acc.setdefault(k, []).append(ms)
serial = (time.perf_counter() - t0) / R
assert bool((ok.cpu() == torch.from_numpy(synth.expected_ok(n))).all())
eng.set_in_flight(2)
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for _ in range(2): call(ok, s1); call(ok2, s2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): call(ok, s1); call(ok2, s2)
torch.cuda.synchronize()
infl = (time.perf_counter() - t0) / 10
assert bool((ok2.cpu() == torch.from_numpy(synth.expected_ok(n))).all())
print(json.dumps({"serial_ms": round(serial * 1e3, 3), "in_flight_ms": round(infl * 1e3, 3), "stages": {k: round(sorted(x)[len(x) // 2], 3) for k, x in acc.items()}, "redo": eng.last_redo_tasks()}))
''' % str(ROOT)
log2n = sys.argv[1] if len(sys.argv) > 1 else "20"
for rep in range(3):
    for mode in ("0", "1"):
        env = dict(os.environ, PLUME_EQ1_SHORT=mode)
        r = subprocess.run([sys.executable, "-c", CHILD, log2n], env=env, capture_output=True, text=True, timeout=900)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print("PLUME_EQ1_SHORT=" + mode, line[0] if line else ("FAILED " + r.stderr[-1500:]), flush=True)
