"""Round 6: host-pointer sign / verify of 2^20 items from page-locked arrays under explicit piece schedules on TWO lanes -- uniform small pieces, now that the lanes' kernels
really run side by side.  Median of 7 calls each; results checked; the same run's device-resident time beside it.  Usage: python r06_host_sched.py sign|verify"""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import zk_nullifier_sig_amd as plume  # noqa: E402
from tests import synth  # noqa: E402
from zk_nullifier_sig_amd import capi  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "sign"
n, K = 1 << 20, 1024
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


def uni(p, first=None):
    s = [first] if first else []
    while sum(s) + p <= 1024:
        s.append(p)
    if sum(s) < 1024:
        s.append(1024 - sum(s))
    return s


SCHEDS = [None, uni(32), uni(48), uni(64), uni(96), uni(128), uni(64, 32), uni(128, 64), [64, 192, 512, 256], [64] * 4 + [128] * 6, [32] * 4 + [64] * 6 + [128] * 4]
if what == "sign":
    os.environ["PLUME_HOST_SIGN_LANES"] = "2"
e = plume.Engine(0)
if what == "sign":
    ref = None
    for sched in SCHEDS:
        if sched: os.environ["PLUME_HOST_SCHEDULE"] = ",".join(str(x * K) for x in sched)
        else: os.environ.pop("PLUME_HOST_SCHEDULE", None)
        tm, tb = med(lambda: e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so))
        if ref is None: ref = np.array(so["s"])
        assert np.array_equal(so["s"], ref) and not so["status"].any()
        print(f"sign 2 lanes {str(sched)[:60]:60s} median {tm:6.2f} ms  best {tb:6.2f}  = {n / tm / 1e3:5.1f} M/s", flush=True)
else:
    sg = e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"])
    v = synth.corrupt_for_verify(1, b, sg)
    want = synth.expected_ok(n)
    vp = {k: capi.pinned_copy(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    okp = capi.pinned_empty(n)
    for sched in SCHEDS:
        if sched: os.environ["PLUME_HOST_SCHEDULE"] = ",".join(str(x * K) for x in sched)
        else: os.environ.pop("PLUME_HOST_SCHEDULE", None)
        tm, tb = med(lambda: e.verify_batch(1, vp["msgs"], pin["off"], vp["pk"], vp["nullifier"], vp["c"], vp["s"], vp["r_point"], vp["hashed_to_curve_r"], out=okp))
        assert np.array_equal(okp, want)
        print(f"verify 2 lanes {str(sched)[:60]:60s} median {tm:6.2f} ms  best {tb:6.2f}  = {n / tm / 1e3:5.1f} M/s", flush=True)
e.close()
