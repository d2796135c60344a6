"""(round 5) PLUME_HOST_TRACE=1 timelines of the host-pointer sign and verify of 2^20 items from page-locked arrays (the library prints them to stderr), with the wall time of
each call and the device-resident serial time of the same batch beside them.  Usage: PLUME_HOST_TRACE=1 [PLUME_HOST_SIGN_LANES=2] python tests/gpu_debug/host_trace.py [log2n]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import zk_nullifier_sig_amd as plume  # noqa: E402
from tests import synth  # noqa: E402
from zk_nullifier_sig_amd import capi  # noqa: E402

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
what = sys.argv[2] if len(sys.argv) > 2 else "both"
eng = plume.Engine(0)
b = synth.sign_inputs(n)
pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
so["status"] = capi.pinned_empty(n)
dev = torch.device("cuda:0")
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
if what in ("both", "sign"):
    for rep in range(3):
        print(f"=== sign call {rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        eng.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so)
        print(f"=== sign wall {1e3 * (time.perf_counter() - t0):.3f} ms", file=sys.stderr, flush=True)
    d = {k: t(b[k]) for k in ("msgs", "sk", "r")}
    off = t(b["off"].view(np.int64))
    o = {k: torch.zeros((n, w), dtype=torch.uint8, device=dev) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
    st = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng.set_stage_timing(True)                     # (off by default since library 0.5; the host-pointer calls above ran without stage events, as a caller's would)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.sign_batch_device(1, n, d["msgs"], off, int(b["off"][-1]), d["sk"], d["r"], None, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], st)
        torch.cuda.synchronize()
        print(f"=== sign device-resident {1e3 * (time.perf_counter() - t0):.3f} ms  {dict((k, round(v, 3)) for k, v in eng.last_stage_times())}", file=sys.stderr, flush=True)
    eng.set_stage_timing(False)
else:
    eng.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so)
if what in ("both", "verify"):
    sg = {k: np.array(so[k]) for k in so}
    v = synth.corrupt_for_verify(1, b, sg)
    vp = {k: capi.pinned_copy(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    ok = capi.pinned_empty(n)
    for rep in range(3):
        print(f"=== verify call {rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        eng.verify_batch(1, vp["msgs"], pin["off"], vp["pk"], vp["nullifier"], vp["c"], vp["s"], vp["r_point"], vp["hashed_to_curve_r"], out=ok)
        print(f"=== verify wall {1e3 * (time.perf_counter() - t0):.3f} ms", file=sys.stderr, flush=True)
    assert np.array_equal(ok, synth.expected_ok(n))
    dv = {k: t(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    off = t(b["off"].view(np.int64))
    okd = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng.set_stage_timing(True)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.verify_batch_device(1, n, dv["msgs"], off, int(b["off"][-1]), dv["pk"], dv["nullifier"], dv["c"], dv["s"], dv["r_point"], dv["hashed_to_curve_r"], okd)
        torch.cuda.synchronize()
        print(f"=== verify device-resident {1e3 * (time.perf_counter() - t0):.3f} ms  {dict((k, round(v, 3)) for k, v in eng.last_stage_times())}", file=sys.stderr, flush=True)
eng.close()
