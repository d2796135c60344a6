"""The signer's uniform schedules (plume_set_sign_uniform levels 1 and 2) against the default one: stage times of a device-resident 2^20 V1 sign."""
import os; os.environ.setdefault("PLUME_STAGE_TIMES", "1")   # the stage-timing events are off by default since library 0.5; this script reads them
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import torch
import zk_nullifier_sig_amd as plume
from tests import synth
n = 1 << 20
b = synth.sign_inputs(n)
e = plume.Engine(0)
dev = torch.device("cuda:0")
d = {k: torch.from_numpy(b[k]).to(dev) for k in ("msgs", "sk", "r")}
off = torch.from_numpy(b["off"].view(np.int64)).to(dev)
o = {k: torch.zeros((n, w), dtype=torch.uint8, device=dev) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
st = torch.zeros(n, dtype=torch.uint8, device=dev)
res = {}
for uni in (0, 1, 2, 0, 1, 2):
    e.set_sign_uniform(uni)
    acc = {}
    for rep in range(4):
        e.sign_batch_device(1, n, d["msgs"], off, int(b["off"][-1]), d["sk"], d["r"], None, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], st)
        torch.cuda.synchronize()
        if rep:
            for name, ms in e.last_stage_times():
                acc[name] = acc.get(name, 0.0) + ms / 3
    print(("default", "level 1", "level 2")[uni], {k: round(v, 3) for k, v in acc.items()}, "total", round(sum(acc.values()), 3), flush=True)
    res[uni] = o["s"].cpu().numpy().copy()
assert np.array_equal(res[1], res[0]) and np.array_equal(res[2], res[0])
