"""Round 3: what crafted items cost the verifier.  A crafted item steers its accumulator into p == +-q inside an unchecked addition (pk = G, s = c = d
small: the first window adds d*G, the pk slot then -d*G).  Rounds 1-2 redid such a lane in place while its wavefront waited; now the task is filed and redone by a second,
dense launch.  Prints ms per 2^20-item V1 verify for: honest, 1 crafted item per 64 (every wavefront of equation 1 holds one), 1 per 8, all crafted; with the number of
redone tasks and a verdict check against the CPU (oracle/plume_cpu_fast.c).

    python tests/gpu_debug/adversarial_timing.py [--log2 20]"""
import os; os.environ.setdefault("PLUME_STAGE_TIMES", "1")   # the stage-timing events are off by default since library 0.5; this script reads them
import argparse
import json
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=20)
    a = ap.parse_args()
    import torch
    import zk_nullifier_sig_amd as plume
    from oracle import plume_oracle as O
    from tests import _cpu_fast as CF
    from tests import synth
    n = 1 << a.log2
    dev = torch.device("cuda:0")
    eng = plume.Engine(0)
    b = synth.sign_inputs(n)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    g = np.frombuffer(O.pt_bytes(O.G), dtype=np.uint8)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    for name, stride in (("honest", 0), ("1 crafted per 64", 64), ("1 crafted per 8", 8), ("all crafted", 1)):
        v = {k: sg[k].copy() for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
        if stride:
            idx = np.arange(7 % stride, n, stride)
            d = (idx % 8 + 1).astype(np.uint8)
            v["pk"][idx] = g
            v["c"][idx] = 0; v["c"][idx, 31] = d
            v["s"][idx] = 0; v["s"][idx, 31] = d
        dv = {k: t(x) for k, x in v.items()}
        msgs, off = t(b["msgs"]), t(b["off"].view(np.int64))
        ok = torch.zeros(n, dtype=torch.uint8, device=dev)
        fn = lambda: eng.verify_batch_device(1, n, msgs, off, int(b["off"][-1]), dv["pk"], dv["nullifier"], dv["c"], dv["s"], dv["r_point"], dv["hashed_to_curve_r"], ok)  # noqa: E731
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        redo = eng.last_redo_tasks()
        st = dict(eng.last_stage_times())
        want = CF.verify_batch(1, b["msgs"], b["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"], nthreads=32)
        same = bool(np.array_equal(ok.cpu().numpy(), want))
        print(json.dumps({"batch": name, "items": n, "verify_ms": round(min(ts), 3), "msm_stage_ms": round(st.get("verify_msm", 0.0), 3), "redone_tasks": redo, "valid": int(want.sum()),
                          "verdicts_equal_cpu": same}), flush=True)
        assert same


if __name__ == "__main__":
    main()
