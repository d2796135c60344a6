"""Round 5: the multi-scalar kernel of small calls as two half chains per task (k_verify_msm_pair; env PLUME_MSM_PAIR_MAX, read at context creation) against one lane per
chain, 2^10 .. 2^16 items, V1 and V2, interleaved on one box; back-to-back calls and single calls with a host wait.    python3 tests/gpu_debug/pair_ab.py"""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import zk_nullifier_sig_amd as plume
from tests import synth

dev = torch.device("cuda:0")
engines = {}
for pair_max in (1 << 20, 0):                 # pair form for every size here / never
    os.environ["PLUME_MSM_PAIR_MAX"] = str(pair_max)
    engines[pair_max] = plume.Engine(0)
os.environ.pop("PLUME_MSM_PAIR_MAX")
print(engines[0].version())
for ver in (1, 2):
    for log2n in (10, 12, 13, 14, 15, 16):
        n = 1 << log2n
        b = synth.sign_inputs(n)
        signed = engines[0].sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
        v = synth.corrupt_for_verify(ver, b, signed)
        t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r") if k in v}
        off = torch.from_numpy(v["off"].view(np.int64)).to(dev)
        mb = int(v["off"][-1])
        exp = synth.expected_ok(n)
        rows = {}
        for rnd in range(3):
            for pm, eng in engines.items():
                ok = torch.zeros(n, dtype=torch.uint8, device=dev)
                call = lambda: eng.verify_batch_device(ver, n, t["msgs"], off, mb, t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"] if ver == 1 else None, t["hashed_to_curve_r"] if ver == 1 else None, ok)  # noqa: E731
                for _ in range(5): call()
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
                rows.setdefault(pm, []).append((b2b * 1e3, float(np.median(lat)) * 1e3))
        a, bb = np.array(rows[1 << 20]), np.array(rows[0])
        print(f"V{ver} 2^{log2n}: two half chains {a[:, 0].min():.4f} ms back to back / {a[:, 1].min():.4f} single;  one lane per chain {bb[:, 0].min():.4f} / {bb[:, 1].min():.4f}   ({100 * (a[:, 0].min() / bb[:, 0].min() - 1):+.1f} %)", flush=True)
e = engines[1 << 20]
e.set_stage_timing(True)
ok = torch.zeros(n, dtype=torch.uint8, device=dev)
