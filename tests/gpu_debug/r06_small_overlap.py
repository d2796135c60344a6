"""Round 6: do small verify calls issued on two streams (two lanes of one context) run side by side?  2^12-item calls leave the chip almost empty, so two in flight should double
the rate; the sweep (profiles/r06_in_flight_small_calls.txt) saw no gain.  Run under `rocprofv3 --kernel-trace`; tests/gpu_debug/r06_l.sh reduces the trace.
    PLUME_IN_FLIGHT_MIN=0 python3 tests/gpu_debug/r06_small_overlap.py [log2n=12] [mode: lanes | engines]"""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import zk_nullifier_sig_amd as plume
from tests import synth

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mode = sys.argv[2] if len(sys.argv) > 2 else "lanes"
dev = torch.device("cuda:0")
n = 1 << log2n
engs = [plume.Engine(0)] if mode == "lanes" else [plume.Engine(0), plume.Engine(0)]
if mode == "lanes":
    engs[0].set_in_flight(2)
b = synth.sign_inputs(n)
signed = engs[0].sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, signed)
t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
off = torch.from_numpy(v["off"].view(np.int64)).to(dev)
oks = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def call(k):
    engs[k % len(engs)].verify_batch_device(1, n, t["msgs"], off, int(v["off"][-1]), t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], oks[k], stream=streams[k])


for _ in range(5):
    call(0); call(1)
torch.cuda.synchronize()
for label, both in (("one stream", False), ("two streams", True), ("one stream", False), ("two streams", True)):
    t0 = time.perf_counter()
    for _ in range(40):
        call(0)
        call(1 if both else 0)
    torch.cuda.synchronize()
    print(f"2^{log2n} {mode}: {label}: {(time.perf_counter() - t0) / 80 * 1e3:.4f} ms per call", flush=True)
assert bool((oks[0].cpu() == torch.from_numpy(synth.expected_ok(n))).all())
