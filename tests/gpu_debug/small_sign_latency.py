"""Round 5: what a small device-resident sign call costs and where (stage events on for the breakdown, off for the time).    python3 tests/gpu_debug/small_sign_latency.py"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import zk_nullifier_sig_amd as plume
from tests import synth
eng = plume.Engine(0)
dev = torch.device("cuda:0")
for log2n in (10, 12, 14, 16):
    n = 1 << log2n
    b = synth.sign_inputs(n)
    d = {k: torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ("msgs", "sk", "r")}
    off = torch.from_numpy(b["off"].view(np.int64)).to(dev)
    o = {k: torch.zeros((n, w), dtype=torch.uint8, device=dev) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
    st = torch.zeros(n, dtype=torch.uint8, device=dev)
    call = lambda: eng.sign_batch_device(1, n, d["msgs"], off, int(b["off"][-1]), d["sk"], d["r"], None, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], st)  # noqa: E731
    for lvl in (1, 0):
        eng.set_sign_uniform(lvl)
        eng.set_stage_timing(False)
        for _ in range(10): call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40): call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 40
        eng.set_stage_timing(True)
        call(); torch.cuda.synchronize()
        print(f"2^{log2n} level {lvl}: {dt * 1e3:.4f} ms per call  {[(k, round(v, 3)) for k, v in eng.last_stage_times()]}", flush=True)
eng.close()
