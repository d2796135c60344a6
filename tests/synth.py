"""Synthetic batches of BASELINE.md §3, vectorised: blk(tag, i) = SHA256(tag || LE64(seed) || LE64(i)),
sk_i = (BE(blk("sk", i)) mod (n-1)) + 1, r_i likewise, m_i = blk("msg", i) (32 bytes).  Pure hashlib/numpy —
shared by tests, smoke() and bench.py (it is input generation, not part of the path under test)."""
import hashlib

import numpy as np

SEED = 0x504C554D45
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def _blk(tag: bytes, i: int, seed: int) -> bytes:
    return hashlib.sha256(tag + seed.to_bytes(8, "little") + i.to_bytes(8, "little")).digest()


def sign_inputs(n: int, start: int = 0, seed: int = SEED):
    sk = bytearray(32 * n)
    r = bytearray(32 * n)
    msg = bytearray(32 * n)
    for k in range(n):
        i = start + k
        sk[32 * k:32 * k + 32] = (int.from_bytes(_blk(b"sk", i, seed), "big") % (N - 1) + 1).to_bytes(32, "big")
        r[32 * k:32 * k + 32] = (int.from_bytes(_blk(b"r", i, seed), "big") % (N - 1) + 1).to_bytes(32, "big")
        msg[32 * k:32 * k + 32] = _blk(b"msg", i, seed)
    off = (np.arange(n + 1, dtype=np.uint64) * 32)
    msgs = np.frombuffer(bytes(msg) + bytes(16), dtype=np.uint8).copy()
    return dict(msgs=msgs, off=off, sk=np.frombuffer(bytes(sk), dtype=np.uint8).reshape(n, 32).copy(),
                r=np.frombuffer(bytes(r), dtype=np.uint8).reshape(n, 32).copy())


def corrupt_for_verify(version: int, b, signed, start: int = 0):
    """verify batch from a signed batch: items with i mod 16 == 5 corrupted by kind (i div 16) mod 4:
    0: s ^= 1 (last byte), 1: c[31] ^= 1, 2: nullifier <- nullifier of item i-1, 3: V1 swap r_point<->hashed_to_curve_r / V2 m[0] ^= 1"""
    n = len(b["off"]) - 1
    v = dict(msgs=b["msgs"].copy(), off=b["off"], pk=signed["pk"].copy(), nullifier=signed["nullifier"].copy(), c=signed["c"].copy(), s=signed["s"].copy())
    if version == 1:
        v["r_point"] = signed["r_point"].copy()
        v["hashed_to_curve_r"] = signed["hashed_to_curve_r"].copy()
    idx = np.arange(n) + start
    bad = np.nonzero(idx % 16 == 5)[0]
    kind = (idx[bad] // 16) % 4
    k0, k1, k2, k3 = bad[kind == 0], bad[kind == 1], bad[kind == 2], bad[kind == 3]
    v["s"][k0, 31] ^= 1
    v["c"][k1, 31] ^= 1
    k2 = k2[k2 > 0]
    v["nullifier"][k2] = signed["nullifier"][k2 - 1]
    if version == 1:
        v["r_point"][k3], v["hashed_to_curve_r"][k3] = signed["hashed_to_curve_r"][k3], signed["r_point"][k3]
    else:
        v["msgs"][(b["off"][k3]).astype(np.int64)] ^= 1
    return v


def expected_ok(n: int, start: int = 0):
    idx = np.arange(n) + start
    return (idx % 16 != 5).astype(np.uint8)
