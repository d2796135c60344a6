"""Host-pointer (PCIe-inclusive) verify / sign at 2^20 items with page-locked caller arrays: one lane (rounds 1-3) against two lanes + four staging slots (round 4),
swept over the piece schedule, next to the device-resident serial rate of the same box."""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import torch
import zk_nullifier_sig_amd as plume
from zk_nullifier_sig_amd import capi
from tests import synth

n = 1 << 20
b = synth.sign_inputs(n)
e0 = plume.Engine(0)
ref = e0.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, ref)
want = synth.expected_ok(n)
pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
vp = {k: capi.pinned_copy(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
so["status"] = capi.pinned_empty(n)
okp = capi.pinned_empty(n)
dev = torch.device("cuda:0")
t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
off = torch.from_numpy(v["off"].view(np.int64)).to(dev)
okd = torch.zeros(n, dtype=torch.uint8, device=dev)


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


def dres():
    e0.verify_batch_device(1, n, t["msgs"], off, int(v["off"][-1]), t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], okd)
    torch.cuda.synchronize()


td = best(dres)
print(f"device-resident serial verify: {td:.2f} ms = {n / td / 1e3:.1f} M/s", flush=True)
e0.close()
for lanes in (1, 2):
    os.environ["PLUME_HOST_LANES"] = str(lanes)
    e = plume.Engine(0)
    for piece, first, tail in [(19, 16, 17), (18, 16, 17), (19, 15, 17), (19, 15, 16), (18, 15, 16), (19, 16, 16)]:
        e.set_host_piece(1 << piece); e.set_host_first_piece(1 << first); e.set_host_tail_piece(1 << tail)
        tv = best(lambda: e.verify_batch(1, vp["msgs"], pin["off"], vp["pk"], vp["nullifier"], vp["c"], vp["s"], vp["r_point"], vp["hashed_to_curve_r"], out=okp))
        assert np.array_equal(okp, want)
        ts = best(lambda: e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so))
        assert np.array_equal(so["s"], ref["s"]) and np.array_equal(so["hashed_to_curve_r"], ref["hashed_to_curve_r"])
        print(f"lanes {lanes} largest 2^{piece} first 2^{first} tail 2^{tail}:  verify {tv:6.2f} ms = {n / tv / 1e3:5.1f} M/s ({td / tv:.3f} of device-resident)   sign {ts:6.2f} ms = {n / ts / 1e3:5.1f} M/s", flush=True)
    e.close()
