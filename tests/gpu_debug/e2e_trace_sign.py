"""One host-pointer SIGN of 2^20 items (page-locked arrays) after two warm-up calls, for a kernel + memory-copy trace of the pipeline:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/e2e_trace_sign -o e2es -- python3 tests/gpu_debug/e2e_trace_sign.py
   python3 tests/gpu_debug/trace_timeline.py gpurun_out/e2e_trace_sign"""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from zk_nullifier_sig_amd import capi
from tests import synth

n = 1 << 20
b = synth.sign_inputs(n)
e = plume.Engine(0)
pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
so["status"] = capi.pinned_empty(n)
for _ in range(3):
    time.sleep(0.05)
    t0 = time.perf_counter()
    e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so)
    print(f"call: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
