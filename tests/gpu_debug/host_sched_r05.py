"""(round 5) host-pointer sign / verify of 2^20 items from page-locked arrays under explicit piece schedules (PLUME_HOST_SCHEDULE), the signer on one lane and on two
(PLUME_HOST_SIGN_LANES, read when the engine is created).  Median of 7 calls each; results checked.  Usage: python host_sched_r05.py sign|verify"""
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import zk_nullifier_sig_amd as plume  # noqa: E402
from tests import synth  # noqa: E402
from zk_nullifier_sig_amd import capi  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "sign"
n = 1 << 20
K = 1024
b = synth.sign_inputs(n)
pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
so["status"] = capi.pinned_empty(n)


def med(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3, min(ts) * 1e3


SIGN = [[64, 192, 512, 192, 64], [64, 128, 192, 192, 192, 128, 96, 32], [128] * 8, [64] + [144] * 6 + [96], [64, 192, 256, 256, 128, 64, 64], [64] * 16, [128, 256, 256, 192, 128, 64],
        [64, 192, 320, 256, 128, 64], [96, 256, 256, 192, 128, 96], [64, 160, 256, 256, 160, 96, 32], [256, 256, 256, 128, 96, 32]]
VERIFY = [[64, 192, 512, 256], [64, 128, 256, 576], [32, 96, 288, 608], [64, 192, 384, 384], [64, 128, 192, 256, 384], [128] * 8, [64, 192, 768]]
if what == "sign":
    ref = None
    for lanes in (1, 2):
        os.environ["PLUME_HOST_SIGN_LANES"] = str(lanes)
        os.environ.pop("PLUME_HOST_SCHEDULE", None)
        e = plume.Engine(0)
        for sched in SIGN:
            assert sum(sched) * K == n, sched
            os.environ["PLUME_HOST_SCHEDULE"] = ",".join(str(x * K) for x in sched)
            tm, tb = med(lambda: e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so))
            if ref is None:
                ref = np.array(so["s"])
            assert np.array_equal(so["s"], ref) and not so["status"].any()
            print(f"sign lanes {lanes} {str(sched):48s} median {tm:6.2f} ms  best {tb:6.2f}  = {n / tm / 1e3:5.1f} M/s", flush=True)
        e.close()
else:
    e = plume.Engine(0)
    sg = e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"])
    v = synth.corrupt_for_verify(1, b, sg)
    want = synth.expected_ok(n)
    vp = {k: capi.pinned_copy(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    okp = capi.pinned_empty(n)
    for lanes in (2, 1):
        e.set_host_lanes(lanes)
        for sched in VERIFY:
            assert sum(sched) * K == n, sched
            os.environ["PLUME_HOST_SCHEDULE"] = ",".join(str(x * K) for x in sched)
            tm, tb = med(lambda: e.verify_batch(1, vp["msgs"], pin["off"], vp["pk"], vp["nullifier"], vp["c"], vp["s"], vp["r_point"], vp["hashed_to_curve_r"], out=okp))
            assert np.array_equal(okp, want)
            print(f"verify lanes {lanes} {str(sched):48s} median {tm:6.2f} ms  best {tb:6.2f}  = {n / tm / 1e3:5.1f} M/s", flush=True)
    e.close()
