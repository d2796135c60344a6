"""Round 6, pricing base-8 Eisenstein digits for equation 2 BEFORE building them: stage times of a 2^20 V1 verify with
  PLUME_HIP_LIB unset                    the shipped kernels
  PLUME_HIP_LIB=.../libplume_hip_b8m.so  -DPLUME_EXP_B8:  equation 2's chain as 44 positions x (3 doublings + 2 additions at 63/64 density) on stand-in digits (garbage verdicts)
  PLUME_HIP_LIB=.../libplume_hip_b8t.so  -DPLUME_EXP_B8T: pass B computes and stores eight more rows for the H and nullifier jobs (stand-in inverses)
Run once per library (the library is chosen at import): tests/gpu_debug/r06_b.sh alternates them on one box."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import zk_nullifier_sig_amd as plume
from tests import synth

dev = torch.device("cuda:0")
eng = plume.Engine(0)
n = 1 << 20
b = synth.sign_inputs(n)
signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, signed)
t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
off = torch.from_numpy(v["off"].view(np.int64)).to(dev)
ok = torch.zeros(n, dtype=torch.uint8, device=dev)
eng.set_stage_timing(True)
call = lambda: eng.verify_batch_device(1, n, t["msgs"], off, int(v["off"][-1]), t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], ok)  # noqa: E731
for _ in range(3):
    call()
torch.cuda.synchronize()
acc, tot = {}, []
for _ in range(6):
    t0 = time.perf_counter(); call(); torch.cuda.synchronize(); tot.append((time.perf_counter() - t0) * 1e3)
    for k, ms in eng.last_stage_times():
        acc.setdefault(k, []).append(ms)
print(eng.version().split("build=")[1], "accepted", int(ok.sum()), "of", n, "| step", round(float(np.median(tot)), 3), "ms |", {k: round(float(np.median(x)), 3) for k, x in acc.items()}, "| clock", eng.last_msm_clock_ghz())
eng.close()
