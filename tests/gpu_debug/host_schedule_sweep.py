"""e2e host-pointer verify / sign at 2^20 under explicit piece schedules (PLUME_HOST_SCHEDULE), two lanes."""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from zk_nullifier_sig_amd import capi
from tests import synth

n = 1 << 20
b = synth.sign_inputs(n)
e = plume.Engine(0)
ref = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, ref)
want = synth.expected_ok(n)
pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
vp = {k: capi.pinned_copy(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
so["status"] = capi.pinned_empty(n)
okp = capi.pinned_empty(n)


def best(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


K = 1024
for sched in [[64, 192, 512, 256], [16, 48, 128, 320, 512], [16, 64, 192, 448, 304], [32, 96, 256, 384, 256], [32, 96, 288, 608], [64, 192, 768], [128, 384, 512], [32, 64, 128, 288, 512],
              [64, 192, 384, 256, 128], [64, 192, 448, 256, 64], [16, 32, 64, 192, 464, 256], [64, 192, 512, 256]]:
    os.environ["PLUME_HOST_SCHEDULE"] = ",".join(str(x * K) for x in sched)
    tv = best(lambda: e.verify_batch(1, vp["msgs"], pin["off"], vp["pk"], vp["nullifier"], vp["c"], vp["s"], vp["r_point"], vp["hashed_to_curve_r"], out=okp))
    assert np.array_equal(okp, want)
    ts = best(lambda: e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so))
    assert np.array_equal(so["s"], ref["s"])
    print(f"{str(sched):40s} verify {tv:6.2f} ms = {n / tv / 1e3:5.1f} M/s   sign {ts:6.2f} ms = {n / ts / 1e3:5.1f} M/s", flush=True)
