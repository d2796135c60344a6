"""Round 6: the signer's host-pointer call on two lanes (uniform 2^16-item pieces, 2^17 beyond 32 pieces) at ragged sizes between 2^18 and 2^21 + 2^17 with the library's
default knobs, from page-locked arrays, V1 and V2, plain and SEC1 outputs: every output byte against the CPU (oracle/plume_cpu_fast.c).  python3 r06_sign_soak.py [calls=30] [seed=1]"""
import os, sys, pathlib, random, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from zk_nullifier_sig_amd import capi
from tests import synth, _cpu_fast as CF

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
T = min(64, os.cpu_count() or 1)
eng = plume.Engine(0)
print(eng.version(), flush=True)
OUT = ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")
checked, t0 = 0, time.time()
for it in range(calls):
    n = rng.choice([(1 << 18), (1 << 18) + 1, (1 << 19) - 1, (1 << 20) + 65537, (1 << 21) + 1, (1 << 21) + (1 << 17)]) if rng.random() < 0.4 else rng.randrange(1 << 18, (1 << 21) + (1 << 17))
    ver = rng.choice([1, 2])
    eng.set_sign_uniform(rng.choice([1, 1, 0, 2]))
    b = synth.sign_inputs(n, start=rng.randrange(1 << 40), seed=rng.randrange(1 << 30))
    pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
    so = {k: capi.pinned_empty((n, 64)) for k in ("pk", "nullifier", "r_point", "hashed_to_curve_r")}
    so.update({k: capi.pinned_empty((n, 32)) for k in ("c", "s")})
    so["status"] = capi.pinned_empty(n)
    so["status"][:] = 0xFF
    got = eng.sign_batch(ver, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so)
    assert not got["status"].any(), (it, n)
    want = CF.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"], nthreads=T)
    for k in OUT:
        assert np.array_equal(got[k], want[k]), (it, n, ver, k, np.nonzero((got[k] != want[k]).any(axis=1))[0][:5])
    checked += n
    print(f"call {it}: {n} items V{ver} ok   ({checked} signatures checked, {time.time() - t0:.0f} s)", flush=True)
eng.close()
print("sign soak ok:", checked, "signatures, every byte")
