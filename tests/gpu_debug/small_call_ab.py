"""Round 5: what a 2^16-item verify call costs with / without (a) the timing events between its kernels (plume_set_stage_timing) and (b) the scalar stage as a launch of
its own instead of role B's last duty in the two-role ingest kernel (env PLUME_SPLIT_SCALARS, read at context creation).  One process, one box, engines side by side;
back-to-back calls (throughput of a stream of small calls) and single calls with a host wait after each (latency).  Also 2^14 and 2^12.
    python3 tests/gpu_debug/small_call_ab.py"""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import zk_nullifier_sig_amd as plume
from tests import synth

dev = torch.device("cuda:0")
engines = {}
for split in (1, 0):
    os.environ["PLUME_SPLIT_SCALARS"] = str(split)
    engines[split] = plume.Engine(0)
os.environ.pop("PLUME_SPLIT_SCALARS")
print(engines[1].version())
for log2n in (16, 14, 12):
    n = 1 << log2n
    b = synth.sign_inputs(n)
    signed = engines[1].sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, signed)
    t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    off = torch.from_numpy(v["off"].view(np.int64)).to(dev)
    mb = int(v["off"][-1])
    exp = synth.expected_ok(n)
    rows = {}
    for rnd in range(3):                                     # interleaved rounds: drifts of the box hit every variant alike
        for split in (1, 0):
            for ev in (0, 1):
                eng = engines[split]
                eng.set_stage_timing(bool(ev))
                ok = torch.zeros(n, dtype=torch.uint8, device=dev)
                call = lambda: eng.verify_batch_device(1, n, t["msgs"], off, mb, t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], ok)  # noqa: E731
                for _ in range(3): call()
                torch.cuda.synchronize()
                reps = 40
                t0 = time.perf_counter()
                for _ in range(reps): call()
                torch.cuda.synchronize()
                b2b = (time.perf_counter() - t0) / reps
                lat = []
                for _ in range(20):
                    t0 = time.perf_counter(); call(); torch.cuda.synchronize(); lat.append(time.perf_counter() - t0)
                assert np.array_equal(ok.cpu().numpy(), exp)
                rows.setdefault((split, ev), []).append((b2b * 1e3, float(np.median(lat)) * 1e3))
    for (split, ev), r in sorted(rows.items(), reverse=True):
        a = np.array(r)
        print(f"2^{log2n}  scalars {'in the ingest kernel' if split else 'as their own launch '}  stage events {'on ' if ev else 'off'} : back to back {a[:, 0].min():.4f} ms (runs {np.round(a[:, 0], 4).tolist()}), single call + wait {a[:, 1].min():.4f} ms")
engines[1].set_stage_timing(True)
ok = torch.zeros(n, dtype=torch.uint8, device=dev)
engines[1].verify_batch_device(1, n, t["msgs"], off, mb, t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], ok); torch.cuda.synchronize()
print("stages of the fused call:", [(k, round(x, 4)) for k, x in engines[1].last_stage_times()])
