"""e2e host-pointer SIGN at 2^20 (page-locked arrays) under explicit piece schedules (PLUME_HOST_SCHEDULE, K items x 1024), one lane; median of 7."""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from zk_nullifier_sig_amd import capi
from tests import synth
n = 1 << 20
b = synth.sign_inputs(n)
e = plume.Engine(0)
e.set_chunk(1 << 20)
ref = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
so["status"] = capi.pinned_empty(n)
def best(fn, reps=7):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3
for sched in [None, [64, 192, 512, 192, 64], [64, 896, 64], [64, 192, 704, 64], [128, 832, 64], [64, 960], [32, 96, 832, 64], [64, 448, 448, 64], [64, 256, 640, 64], [64, 896, 32, 32], [1024]]:
    if sched: os.environ["PLUME_HOST_SCHEDULE"] = ",".join(str(x * 1024) for x in sched)
    t = best(lambda: e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so))
    assert np.array_equal(so["s"], ref["s"]) and np.array_equal(so["nullifier"], ref["nullifier"])
    print(f"{str(sched):32s} sign {t:6.2f} ms = {n / t / 1e3:5.1f} M/s", flush=True)
