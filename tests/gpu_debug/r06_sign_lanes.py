"""Round 6: the signer's host-pointer call from page-locked arrays, one lane (tapered pieces, the round-5 default) against two lanes (uniform 2^16-item pieces, the round-6
default), at several batch sizes.  PLUME_HOST_SIGN_LANES is read when the context is created: one process per setting.  Usage: python r06_sign_lanes.py <log2 n>..."""
import os, sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import zk_nullifier_sig_amd as plume  # noqa: E402
from tests import synth  # noqa: E402
from zk_nullifier_sig_amd import capi  # noqa: E402

e = plume.Engine(0)
for lg in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << lg
    b = synth.sign_inputs(n)
    pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
    so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
    so["status"] = capi.pinned_empty(n)
    call = lambda: e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so)
    call()
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
    assert not so["status"].any()
    import hashlib
    h = hashlib.sha256(so["s"].tobytes() + so["nullifier"].tobytes()).hexdigest()[:16]
    tm = sorted(ts)[len(ts) // 2] * 1e3
    print(f"sign lanes={os.environ.get('PLUME_HOST_SIGN_LANES', 'default')} 2^{lg}: median {tm:7.2f} ms  best {min(ts) * 1e3:7.2f}  = {n / tm / 1e3:5.1f} M/s  digest {h}", flush=True)
