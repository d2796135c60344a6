"""Host-pointer (PCIe-inclusive) API timing at 2^20 items with page-locked caller arrays, swept over the piece schedule
(largest / first / tail piece), plus pageable arrays, per-call registration, and a two-shard context on one GPU."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from zk_nullifier_sig_amd import capi
from tests import synth

n = 1 << 20
b = synth.sign_inputs(n)
eng = plume.Engine(0)
ref = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, ref)
want = synth.expected_ok(n)
pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
vp = {k: capi.pinned_copy(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
so["status"] = capi.pinned_empty(n)
okp = capi.pinned_empty(n)


def best(fn, reps=4):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


def run(e, label):
    tv = best(lambda: e.verify_batch(1, vp["msgs"], pin["off"], vp["pk"], vp["nullifier"], vp["c"], vp["s"], vp["r_point"], vp["hashed_to_curve_r"], out=okp))
    assert np.array_equal(okp, want)
    ts = best(lambda: e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so))
    assert np.array_equal(so["s"], ref["s"])
    print(f"{label:58s} verify {tv:6.2f} ms = {n / tv / 1e3:5.1f} M/s   sign {ts:6.2f} ms = {n / ts / 1e3:5.1f} M/s", flush=True)


for piece, first, tail in [(19, 16, 17), (19, 15, 17), (19, 17, 17), (18, 16, 17), (18, 17, 17), (20, 16, 17), (19, 16, 16), (19, 16, 18), (18, 18, 18), (20, 20, 20)]:
    eng.set_host_piece(1 << piece); eng.set_host_first_piece(1 << first); eng.set_host_tail_piece(1 << tail)
    run(eng, f"pinned, largest 2^{piece} first 2^{first} tail 2^{tail}")
eng.set_host_piece(1 << 19); eng.set_host_first_piece(1 << 16); eng.set_host_tail_piece(1 << 17)
tv = best(lambda: eng.verify_batch(1, v["msgs"], b["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"]))
ts = best(lambda: eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"]))
print(f"{'pageable numpy arrays (default schedule)':58s} verify {tv:6.2f} ms = {n / tv / 1e3:5.1f} M/s   sign {ts:6.2f} ms = {n / ts / 1e3:5.1f} M/s (includes allocating the outputs)")
eng.set_host_register_min(1 << 20)
tv = best(lambda: eng.verify_batch(1, v["msgs"], b["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"]))
print(f"{'pageable + per-call hipHostRegister':58s} verify {tv:6.2f} ms = {n / tv / 1e3:5.1f} M/s")
eng.set_host_register_min(0)
for g in (2, 3):
    m = plume.Engine([0] * g)
    run(m, f"pinned, {g} shards on one GPU (default schedule)")
    m.close()
