"""Round 3: sub-batch overlap A/B on ONE box.  For every zk-nullifier-sig_amd/libplume_hip*.so build and every sub-batch count, time the
device-resident 2^20 V1 verify and V1 sign (same resident batch), check the verdicts / signatures, and print one JSON line per (library, count).

    python tests/gpu_debug/overlap_sweep.py [--log2 20] [--subs 1,2,4,8,16]
Boxes of the pool differ by +-5 %, so only numbers from one invocation are comparable; the default build runs first and again last (drift)."""
import os; os.environ.setdefault("PLUME_STAGE_TIMES", "1")   # the stage-timing events are off by default since library 0.5; this script reads them
import json
import os
import pathlib
import subprocess
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402


def child(lib, data, subs):
    os.environ["PLUME_HIP_LIB"] = lib
    import torch
    import zk_nullifier_sig_amd as plume
    d = np.load(data)
    eng = plume.Engine(0)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(d[k]).to(dev) for k in d.files}
    n = int(t["off"].numel() - 1)
    mb = int(d["off"][-1])
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    o = {k: torch.zeros((n, w), dtype=torch.uint8, device=dev) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
    status = torch.zeros(n, dtype=torch.uint8, device=dev)

    acc_stages = {}

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        ts = []
        acc_stages.clear()
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
            for name, ms in eng.last_stage_times():          # mean over the timed calls (same-stream sub-batches repeat the stage names: summed per call)
                acc_stages[name] = acc_stages.get(name, 0.0) + ms / reps
        # back to back (what bench.py times)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return min(ts), (time.perf_counter() - t0) * 1e3 / reps

    def stages():
        return dict(acc_stages)

    for k in subs:
        eng.set_sub_batches(k)
        ok.zero_()
        v = timed(lambda: eng.verify_batch_device(1, n, t["msgs"], t["off"], mb, t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], ok), 10)
        vst = stages()
        assert bool((ok.cpu().numpy() == d["expected"]).all()), "wrong verdicts"
        o["nullifier"].zero_()
        s = timed(lambda: eng.sign_batch_device(1, n, t["msgs"], t["off"], mb, t["sk"], t["r"], None, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], status), 8)
        sst = stages()
        assert bool((o["nullifier"].cpu().numpy() == d["nullifier_signed"]).all()) and bool((o["s"].cpu().numpy() == d["s_signed"]).all()), "wrong signatures"
        print(json.dumps({"lib": pathlib.Path(lib).name, "sub_batches": k, "verify_best_ms": round(v[0], 3), "verify_b2b_ms": round(v[1], 3), "sign_best_ms": round(s[0], 3),
                          "sign_b2b_ms": round(s[1], 3), "verify_stages": {a: round(b, 3) for a, b in vst.items()}, "sign_stages": {a: round(b, 3) for a, b in sst.items()}}), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], sys.argv[3], [int(x) for x in sys.argv[4].split(",")])
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", type=int, default=20)
    ap.add_argument("--subs", default="1,2,4,8,16")
    a = ap.parse_args()
    import zk_nullifier_sig_amd as plume
    from tests import synth
    n = 1 << a.log2
    b = synth.sign_inputs(n)
    eng = plume.Engine(0)
    signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, signed)
    data = "/tmp/plume_overlap.npz"
    np.savez(data, msgs=v["msgs"], off=v["off"].view(np.int64), pk=v["pk"], nullifier=v["nullifier"], c=v["c"], s=v["s"], r_point=v["r_point"], hashed_to_curve_r=v["hashed_to_curve_r"],
             sk=b["sk"], r=b["r"], expected=synth.expected_ok(n), nullifier_signed=signed["nullifier"], s_signed=signed["s"])
    eng.close()
    libs = sorted((ROOT / "zk-nullifier-sig_amd").glob("libplume_hip*.so"))
    for lib in libs + libs[:1]:
        subprocess.run([sys.executable, __file__, "--child", str(lib), data, a.subs], check=False)


if __name__ == "__main__":
    main()
