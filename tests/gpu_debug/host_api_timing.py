import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from tests import synth
eng = plume.Engine(0)
n = 1 << 20
b = synth.sign_inputs(n)
t0 = time.perf_counter(); signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"]); t1 = time.perf_counter()
print(f"host-pointer sign_batch 2^20 (first call, allocations included): {1e3*(t1-t0):.1f} ms")
for rep in range(3):
    t0 = time.perf_counter(); signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"]); t1 = time.perf_counter()
    print(f"host-pointer sign_batch 2^20: {1e3*(t1-t0):.1f} ms  -> {n/(t1-t0)/1e6:.1f} M/s")
v = synth.corrupt_for_verify(1, b, signed)
for rep in range(4):
    t0 = time.perf_counter(); ok = eng.verify_batch(1, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"]); t1 = time.perf_counter()
    print(f"host-pointer verify_batch 2^20: {1e3*(t1-t0):.1f} ms  -> {n/(t1-t0)/1e6:.1f} M/s")
