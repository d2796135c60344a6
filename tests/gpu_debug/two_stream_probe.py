"""does running two half-batches on two streams (two contexts = two workspaces) beat one full batch? (kernel overlap probe)"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import zk_nullifier_sig_amd as plume
from tests import synth
dev = torch.device("cuda:0")
e1, e2 = plume.Engine(0), plume.Engine(0)
n = 1 << 20; h = n // 2
b = synth.sign_inputs(n)
signed = e1.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
d = {k: t(signed[k]) for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
msgs = t(b["msgs"]); off = t(b["off"].view(np.int64))
off2 = t((b["off"][h:] - b["off"][h]).view(np.int64)); msgs2 = msgs[32 * h:]
ok = torch.zeros(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def full():
    e1.verify_batch_device(1, n, msgs, off, 32 * n, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], ok)
def halves():
    e1.verify_batch_device(1, h, msgs, off, 32 * h, d["pk"][:h], d["nullifier"][:h], d["c"][:h], d["s"][:h], d["r_point"][:h], d["hashed_to_curve_r"][:h], ok[:h], stream=s1)
    e2.verify_batch_device(1, h, msgs2, off2, 32 * h, d["pk"][h:], d["nullifier"][h:], d["c"][h:], d["s"][h:], d["r_point"][h:], d["hashed_to_curve_r"][h:], ok[h:], stream=s2)
for name, fn in (("one stream, 2^20", full), ("two streams, 2 x 2^19", halves), ("one stream, 2^20", full), ("two streams, 2 x 2^19", halves)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    print(f"{name:24s} {dt*1e3:.3f} ms  ok={bool(ok.all())}")
