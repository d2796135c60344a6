"""Round 6: host-pointer verify of 2^20 items from page-locked arrays, two lanes, growth schedules that start below 2^16 items (the first upload is the exposed one).
Median of 9 calls each, interleaved twice; verdicts checked.  Usage: python r06_verify_sched.py"""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import zk_nullifier_sig_amd as plume  # noqa: E402
from tests import synth  # noqa: E402
from zk_nullifier_sig_amd import capi  # noqa: E402

n, K = 1 << 20, 1024
b = synth.sign_inputs(n)
e = plume.Engine(0)
sg = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, sg)
want = synth.expected_ok(n)
vp = {k: capi.pinned_copy(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
off = capi.pinned_copy(b["off"])
okp = capi.pinned_empty(n)
SCHEDS = [None, [64, 192, 512, 256], [32, 96, 288, 512, 96], [32, 64, 192, 512, 224], [32, 128, 384, 480], [48, 144, 432, 400], [32, 160, 512, 320], [16, 48, 144, 432, 384], [64, 192, 384, 384], [64, 256, 512, 192]]
for rep in range(2):
    for sched in SCHEDS:
        if sched:
            assert sum(sched) == 1024
            os.environ["PLUME_HOST_SCHEDULE"] = ",".join(str(x * K) for x in sched)
        else:
            os.environ.pop("PLUME_HOST_SCHEDULE", None)
        call = lambda: e.verify_batch(1, vp["msgs"], off, vp["pk"], vp["nullifier"], vp["c"], vp["s"], vp["r_point"], vp["hashed_to_curve_r"], out=okp)
        call()
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
        assert np.array_equal(okp, want)
        print(f"verify 2 lanes {str(sched):34s} median {sorted(ts)[4] * 1e3:6.2f} ms  best {min(ts) * 1e3:6.2f}", flush=True)
