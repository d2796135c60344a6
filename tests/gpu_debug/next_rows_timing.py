"""Device-resident timings at 2^20 of the entry points that are not on the headline metric's path: verify_non_zk (V1, V2), h2c intermediates,
SEC1-DER scalar export, register packing."""
import os; os.environ.setdefault("PLUME_STAGE_TIMES", "1")   # the stage-timing events are off by default since library 0.5; this script reads them
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import ctypes as C
import numpy as np
import torch
import zk_nullifier_sig_amd as plume
from tests import synth

n = 1 << 20
eng = plume.Engine(0)
dev = torch.device("cuda:0")
b = synth.sign_inputs(n)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d = eng._dp


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


msgs, off = t(b["msgs"]), t(b["off"].view(np.int64))
for ver in (1, 2):
    sg = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    g = {k: t(sg[k]) for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    ms = timed(lambda: eng.verify_non_zk_batch_device(ver, n, msgs, off, 32 * n, g["pk"], g["nullifier"], g["s"], g["r_point"], g["hashed_to_curve_r"], g["c"], ok))
    assert bool((ok == 1).all())
    print(f"verify_non_zk V{ver}: {ms:.2f} ms = {n / ms / 1e3:.1f} M/s", dict(eng.last_stage_times()))
u, mp_, q, h = (torch.zeros((n, w), dtype=torch.uint8, device=dev) for w in (64, 128, 128, 64))
lib = eng._lib
for regs in (0, 1):
    ms = timed(lambda: eng._chk(lib.plume_h2c_intermediates_batch_device(eng._ctx, n, d(msgs), d(off), 32 * n, d(g["pk"]), regs, d(u), d(mp_), d(q), d(h), None), "h2c_intermediates"))
    print(f"h2c_intermediates (registers={regs}): {ms:.2f} ms = {n / ms / 1e3:.1f} M/s")
der = torch.zeros((n, 109), dtype=torch.uint8, device=dev); st = torch.zeros(n, dtype=torch.uint8, device=dev)
sk = t(b["sk"])
ms = timed(lambda: eng._chk(lib.plume_scalars_to_sec1_der_batch_device(eng._ctx, n, d(sk), d(der), d(st), None), "der"))
print(f"scalars_to_sec1_der: {ms:.2f} ms = {n / ms / 1e3:.1f} M/s")
regs = torch.zeros((6 * n, 4), dtype=torch.int64, device=dev)
vals = torch.cat([g["c"], g["s"], g["pk"].reshape(-1, 32), g["nullifier"].reshape(-1, 32)])
ms = timed(lambda: eng._chk(lib.plume_registers_from_be_device(eng._ctx, 6 * n, d(vals), d(regs), None), "regs"))
print(f"registers_from_be (c, s, pk, nullifier of 2^20 signatures = 6 x 2^20 values): {ms:.3f} ms")
