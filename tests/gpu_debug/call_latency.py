"""Wall time of device-resident verify calls: one isolated call vs. calls queued back to back (is there a per-call gap?)."""
import os; os.environ.setdefault("PLUME_STAGE_TIMES", "1")   # the stage-timing events are off by default since library 0.5; this script reads them
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import zk_nullifier_sig_amd as plume
from tests import synth
eng = plume.Engine(0)
dev = torch.device("cuda:0")
n = 1 << 20
b = synth.sign_inputs(n)
signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, signed)
t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "off", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
ok = torch.zeros(n, dtype=torch.uint8, device=dev)
mb = int(v["off"][-1])
def call():
    eng.verify_batch_device(1, n, t["msgs"], t["off"], mb, t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], ok)
for _ in range(2): call()
torch.cuda.synchronize()
for gap in (0.0, 0.05, 0.5):
    ts = []
    for rep in range(4):
        time.sleep(gap)
        t0 = time.perf_counter(); call(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t0), sum(ms for _, ms in eng.last_stage_times())))
    print(f"idle {gap*1e3:5.0f} ms before each call: submit {np.mean([a for a,_,_ in ts]):6.2f} ms, wall {np.mean([b for _,b,_ in ts]):6.2f} ms, GPU stages {np.mean([c for _,_,c in ts]):6.2f} ms")
t0 = time.perf_counter()
for _ in range(8): call()
torch.cuda.synchronize()
print(f"8 calls back to back: {1e3*(time.perf_counter()-t0)/8:6.2f} ms per call")
t0 = time.perf_counter(); e2 = plume.Engine(0); t1 = time.perf_counter()
print(f"plume_init (second context on the device: the generator's fixed tables are shared): {1e3*(t1-t0):.1f} ms")
