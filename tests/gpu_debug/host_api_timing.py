"""Host-pointer (PCIe-inclusive) API timing at 2^20 items, swept over the pipelined piece size (plume_set_host_piece)."""
import sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from tests import synth
eng = plume.Engine(0)
n = 1 << 20
b = synth.sign_inputs(n)
ref_signed = None
for lg in (20, 19, 18, 17, 16):
    eng.set_host_piece(1 << lg)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter(); signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"]); t1 = time.perf_counter()
        if rep: best = min(best, t1 - t0)
    if ref_signed is None: ref_signed = signed
    same = all(np.array_equal(ref_signed[k], signed[k]) for k in ref_signed)
    print(f"sign_batch   2^20, piece 2^{lg}: {1e3*best:7.1f} ms -> {n/best/1e6:5.1f} M/s   identical={same}")
v = synth.corrupt_for_verify(1, b, ref_signed)
want = synth.expected_ok(n)
for lg in (20, 19, 18, 17, 16):
    eng.set_host_piece(1 << lg)
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter(); ok = eng.verify_batch(1, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"]); t1 = time.perf_counter()
        if rep: best = min(best, t1 - t0)
    print(f"verify_batch 2^20, piece 2^{lg}: {1e3*best:7.1f} ms -> {n/best/1e6:5.1f} M/s   correct={np.array_equal(np.asarray(ok).astype(np.uint8), want)}")
