"""Ragged soak (round 5): many small calls of odd sizes through contexts with randomly chosen pipeline settings -- piece sizes, chunk, lanes, sub-batches, the first
equation's form, the signer's level, one to five shards sharing the GPU -- sign on the GPU, compare every output with the CPU (oracle/plume_cpu_fast.c), mutate, verify
in both verify semantics from pageable and page-locked arrays, compare every verdict.    python3 tests/gpu_debug/soak_ragged.py [iterations=300] [seed=1]"""
import os, sys, pathlib, random, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from zk_nullifier_sig_amd import capi
from tests import synth, _fuzz, _cpu_fast as CF

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
T = min(32, os.cpu_count() or 1)
engines = {}
def engine(shards):
    if shards not in engines:
        engines[shards] = plume.Engine(0) if shards == 1 else plume.Engine([0] * shards)
    return engines[shards]
checked = 0
t0 = time.time()
sizes = [1, 2, 3, 5, 63, 64, 65, 127, 255, 256, 257, 1000, 1023, 1025, 4095, 4097, 10000, 65535, 65537]
for it in range(iters):
    n = rng.choice(sizes) if rng.random() < 0.6 else rng.randrange(1, 30000)
    shards = rng.choice([1, 1, 1, 2, 3, 5])
    eng = engine(shards)
    eng.set_host_piece(rng.choice([1 << 19, 4096, 1000, 333, 64]))
    eng.set_host_first_piece(rng.choice([1 << 16, 512, 100, 7]))
    eng.set_host_tail_piece(rng.choice([1 << 16, 256, 50]))
    eng.set_chunk(rng.choice([1 << 20, 8192, 2048]))
    eng.set_eq1_short(rng.choice([1, 3, 3, 0, 2]))
    eng.set_sign_uniform(rng.choice([1, 1, 0, 2]))
    eng.set_host_lanes(rng.choice([1, 2]))
    ver = rng.choice([1, 2])
    b = synth.sign_inputs(n, start=rng.randrange(1 << 40), seed=rng.randrange(1 << 30))
    pinned = rng.random() < 0.5
    arr = (lambda x: capi.pinned_copy(x)) if pinned else (lambda x: x)
    signed = eng.sign_batch(ver, arr(b["msgs"]), arr(b["off"]), arr(b["sk"]), arr(b["r"]))
    want_s = CF.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"], nthreads=T)
    for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r"):
        assert np.array_equal(signed[k], want_s[k]), (it, n, shards, ver, k)
    v = _fuzz.fuzz_verify_batch(ver, signed, b, seed=it)
    rp, hr = (v["r_point"], v["hashed_to_curve_r"]) if ver == 1 else (None, None)
    want = CF.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], rp, hr, nthreads=T)
    got = eng.verify_batch(ver, arr(v["msgs"]), arr(v["off"]), arr(v["pk"]), arr(v["nullifier"]), arr(v["c"]), arr(v["s"]), arr(rp) if ver == 1 else None, arr(hr) if ver == 1 else None)
    assert np.array_equal(got, want), (it, n, shards, ver, "verify", np.nonzero(got != want)[0][:8])
    z = _fuzz.fuzz_non_zk_batch(ver, signed, b, seed=it + 7)
    want_z = CF.verify_non_zk_batch(ver, z["msgs"], z["off"], z["pk"], z["nullifier"], z["s"], z["r_point"], z["hashed_to_curve_r"], z["c"], nthreads=T)
    got_z = eng.verify_non_zk_batch(ver, arr(z["msgs"]), arr(z["off"]), arr(z["pk"]), arr(z["nullifier"]), arr(z["s"]), arr(z["r_point"]), arr(z["hashed_to_curve_r"]), arr(z["c"]))
    assert np.array_equal(got_z, want_z), (it, n, shards, ver, "non_zk", np.nonzero(got_z != want_z)[0][:8])
    checked += 3 * n
    if it % 25 == 24:
        print(f"{it + 1} calls x 3, {checked} item checks, {time.time() - t0:.0f} s", flush=True)
for e in engines.values():
    e.close()
print("ragged soak ok:", iters, "iterations,", checked, "item checks")
