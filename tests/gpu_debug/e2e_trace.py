"""One host-pointer verify of 2^20 items (page-locked arrays) after two warm-up calls, for a kernel + memory-copy trace of the pipeline:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/e2e_trace -o e2e -- python3 tests/gpu_debug/e2e_trace.py"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from zk_nullifier_sig_amd import capi
from tests import synth

n = 1 << 20
b = synth.sign_inputs(n)
e = plume.Engine(0)
ref = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = synth.corrupt_for_verify(1, b, ref)
vp = {k: capi.pinned_copy(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
off = capi.pinned_copy(b["off"])
okp = capi.pinned_empty(n)
for _ in range(3):
    time.sleep(0.05)
    t0 = time.perf_counter()
    e.verify_batch(1, vp["msgs"], off, vp["pk"], vp["nullifier"], vp["c"], vp["s"], vp["r_point"], vp["hashed_to_curve_r"], out=okp)
    print(f"call: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
assert np.array_equal(okp, synth.expected_ok(n))
