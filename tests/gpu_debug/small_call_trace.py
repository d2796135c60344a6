"""Twenty device-resident 2^16-item verify calls back to back, for a kernel trace of one call's launches and the gaps between them:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/small_call -o sc -- python3 tests/gpu_debug/small_call_trace.py
   python3 tests/gpu_debug/trace_timeline.py gpurun_out/small_call"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import zk_nullifier_sig_amd as plume
from tests import synth
eng = plume.Engine(0)
dev = torch.device("cuda:0")
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 16)
b = synth.sign_inputs(n)
signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, signed)
t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "off", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
ok = torch.zeros(n, dtype=torch.uint8, device=dev)
mb = int(v["off"][-1])
def call():
    eng.verify_batch_device(1, n, t["msgs"], t["off"], mb, t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], ok)
for _ in range(3): call()
torch.cuda.synchronize()
time.sleep(0.1)
t0 = time.perf_counter()
for _ in range(20): call()
torch.cuda.synchronize()
print(f"20 calls back to back: {1e3 * (time.perf_counter() - t0) / 20:6.3f} ms per call")
