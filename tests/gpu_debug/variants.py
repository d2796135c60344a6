"""A/B/C... on ONE box: time every zk-nullifier-sig_amd/libplume_hip*.so build on the same device-resident 2^20 batch.

    python tests/gpu_debug/variants.py            (parent: makes the batch with the default build, then one child per library)
Boxes of the pool differ by +-5 %, so only numbers from one invocation are comparable."""
import os; os.environ.setdefault("PLUME_STAGE_TIMES", "1")   # the stage-timing events are off by default since library 0.5; this script reads them
import os, sys, subprocess, pathlib, json, time
ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np


def child(lib, data):
    os.environ["PLUME_HIP_LIB"] = lib
    import torch
    import zk_nullifier_sig_amd as plume
    d = np.load(data)
    eng = plume.Engine(0)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(d[k]).to(dev) for k in d.files}
    n = int(t["off"].numel() - 1)
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    out = {}
    for ver in (1,):
        best = None
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.verify_batch_device(ver, n, t["msgs"], t["off"], int(d["off"][-1]), t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], ok)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            st = dict(eng.last_stage_times())
            if rep and (best is None or dt < best[0]):
                best = (dt, st)
        assert bool((ok.cpu().numpy() == d["expected"]).all()), "wrong verdicts"
        out["verify_ms"] = round(best[0], 3)
        out["stages"] = {k: round(v, 3) for k, v in best[1].items()}
    # sign
    o = {k: torch.zeros((n, w), dtype=torch.uint8, device=dev) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
    status = torch.zeros(n, dtype=torch.uint8, device=dev)
    best = None
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sign_batch_device(1, n, t["msgs"], t["off"], int(d["off"][-1]), t["sk"], t["r"], None, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], status)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        st = dict(eng.last_stage_times())
        if rep and (best is None or dt < best[0]):
            best = (dt, st)
    assert bool((o["nullifier"].cpu().numpy().reshape(n, 64) == d["nullifier_signed"].reshape(n, 64)).all()), "wrong signatures"
    out["sign_ms"] = round(best[0], 3)
    out["sign_stages"] = {k: round(v, 3) for k, v in best[1].items()}
    print(pathlib.Path(lib).name, json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], sys.argv[3])
    import zk_nullifier_sig_amd as plume
    from tests import synth
    lg = int(os.environ.get("LOG2", "20"))
    n = 1 << lg
    b = synth.sign_inputs(n)
    eng = plume.Engine(0)
    signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, signed)
    data = "/tmp/plume_variants.npz"
    np.savez(data, msgs=v["msgs"], off=v["off"], pk=v["pk"], nullifier=v["nullifier"], c=v["c"], s=v["s"], r_point=v["r_point"], hashed_to_curve_r=v["hashed_to_curve_r"],
             sk=b["sk"], r=b["r"], expected=synth.expected_ok(n), nullifier_signed=signed["nullifier"])
    del eng
    libs = sorted((ROOT / "zk-nullifier-sig_amd").glob("libplume_hip*.so"))
    order = libs + libs[:1]          # the default build again at the end: drift check
    for lib in order:
        subprocess.run([sys.executable, __file__, "--child", str(lib), data], check=False)


if __name__ == "__main__":
    main()
