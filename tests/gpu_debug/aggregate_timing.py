"""Stage times of the aggregate pre-filter at 2^20 (and smaller), device-resident, next to the per-item verify on the same batch."""
import os; os.environ.setdefault("PLUME_STAGE_TIMES", "1")   # the stage-timing events are off by default since library 0.5; this script reads them
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import zk_nullifier_sig_amd as plume  # noqa: E402
from tests import synth  # noqa: E402

KEYS = ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")
eng = plume.Engine(0)
dev = "cuda:0"
for lg in (int(a) for a in (sys.argv[1:] or ["20", "16"])):
    n = 1 << lg
    b = synth.sign_inputs(n, start=9_000_000)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    t = {k: torch.from_numpy(sg[k]).to(dev) for k in KEYS}
    msgs = torch.from_numpy(b["msgs"]).to(dev)
    off = torch.from_numpy(b["off"].astype(np.int64)).to(dev)
    res = torch.zeros(72, dtype=torch.uint8, device=dev)
    hok = torch.zeros(n, dtype=torch.uint8, device=dev)
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    seed = bytes(range(32))
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.aggregate_check_device(1, 0, n, msgs, off, 32 * n, *(t[k] for k in KEYS), seed, 0, hok, res)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = eng.last_stage_times()
    rec = res.cpu().numpy()
    print(f"n=2^{lg} aggregate: {dt * 1e3:.2f} ms  ({n / dt / 1e6:.1f} M items/s)  all_ok={rec[0]}", {k: round(v, 3) for k, v in st})
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.verify_batch_device(1, n, msgs, off, 32 * n, t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], ok)
        torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print(f"n=2^{lg} verify:    {dt2 * 1e3:.2f} ms  ({n / dt2 / 1e6:.1f} M items/s)  ratio {dt2 / dt:.2f}x")
