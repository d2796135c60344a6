"""e2e host-pointer SIGN at 2^20 (page-locked arrays): lanes x piece schedule, median of 9."""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from zk_nullifier_sig_amd import capi
from tests import synth
n = 1 << 20
b = synth.sign_inputs(n)
pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
so["status"] = capi.pinned_empty(n)
def best(fn, reps=9):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3, min(ts) * 1e3
for lanes in (1, 2):
    os.environ["PLUME_HOST_LANES"] = str(lanes)
    e = plume.Engine(0)
    for piece, first, tail in [(19, 16, 16), (19, 16, 17), (18, 16, 17), (18, 16, 16), (17, 16, 16), (18, 17, 17), (18, 15, 15)]:
        e.set_host_piece(1 << piece); e.set_host_first_piece(1 << first); e.set_host_tail_piece(1 << tail)
        m, lo = best(lambda: e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so))
        print(f"lanes {lanes} largest 2^{piece} first 2^{first} tail 2^{tail}: sign median {m:6.2f} ms best {lo:6.2f} ms = {n / m / 1e3:5.1f} M/s", flush=True)
    e.close()
