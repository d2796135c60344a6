"""seeded fuzz batches for verify: honest signatures with random field-level mutations, random garbage records,
boundary scalars — expected results come from the C oracle, never from assumptions"""
import random

import numpy as np

from tests import synth

N = synth.N
P = 2**256 - 2**32 - 977


def fuzz_verify_batch(ver, signed, b, seed):
    """signed: output of a sign pass over batch b (numpy arrays).  Returns the mutated verify arrays."""
    rng = random.Random(seed)
    n = len(b["off"]) - 1
    v = dict(msgs=b["msgs"].copy(), off=b["off"], pk=signed["pk"].copy(), nullifier=signed["nullifier"].copy(), c=signed["c"].copy(), s=signed["s"].copy(),
             r_point=signed["r_point"].copy(), hashed_to_curve_r=signed["hashed_to_curve_r"].copy())
    pts = ["pk", "nullifier", "r_point", "hashed_to_curve_r"]
    for i in range(n):
        k = rng.randrange(16)
        if k < 5:
            continue                                   # honest
        f = rng.choice(pts + ["c", "s", "msg"])
        if k == 5:                                     # random bit flip somewhere
            if f == "msg":
                lo, hi = int(b["off"][i]), int(b["off"][i + 1])
                if hi > lo:
                    v["msgs"][rng.randrange(lo, hi)] ^= 1 << rng.randrange(8)
            else:
                w = v[f].shape[1]
                v[f][i, rng.randrange(w)] ^= 1 << rng.randrange(8)
        elif k == 6:                                   # random garbage record
            if f != "msg":
                v[f][i] = np.frombuffer(rng.randbytes(v[f].shape[1]), dtype=np.uint8)
        elif k == 7:                                   # identity
            f = rng.choice(pts)
            v[f][i] = 0
        elif k == 8:                                   # boundary scalars
            f = rng.choice(["c", "s"])
            val = rng.choice([0, 1, N - 1, N, N + 1, 2**256 - 1, 2**255])
            v[f][i] = np.frombuffer(val.to_bytes(32, "big"), dtype=np.uint8)
        elif k == 9:                                   # negate a point (still on the curve)
            f = rng.choice(pts)
            y = int.from_bytes(v[f][i, 32:].tobytes(), "big")
            if y:
                v[f][i, 32:] = np.frombuffer(((P - y) % P).to_bytes(32, "big"), dtype=np.uint8)
        elif k == 10:                                  # non-canonical coordinate (x + p when it fits)
            f = rng.choice(pts)
            x = int.from_bytes(v[f][i, :32].tobytes(), "big")
            if x + P < 2**256:
                v[f][i, :32] = np.frombuffer((x + P).to_bytes(32, "big"), dtype=np.uint8)
        elif k == 11:                                  # fields of another item
            j = rng.randrange(n)
            f = rng.choice(pts + ["c", "s"])
            v[f][i] = v[f][j]
        elif k == 12:                                  # swap R and Hr / pk and nullifier
            a, bb = rng.choice([("r_point", "hashed_to_curve_r"), ("pk", "nullifier")])
            t = v[a][i].copy(); v[a][i] = v[bb][i]; v[bb][i] = t
        # 13..15: honest
    plant_nonce_zero(ver, v, b, seed)
    return v


def plant_nonce_zero(ver, v, b, seed, every=48, tamper=True):
    """Where the batch carries its secret keys (b["sk"]): about one item in `every` becomes the signature with nonce r = 0 -- R = Hr = identity, c = SHA256(.. || 00 || 00)
    mod n, s = c sk -- which the reference's verify ACCEPTS (no nonce check anywhere in rust-k256/src/lib.rs:93-145; R' and Hr' come out as the identity), and half of the
    planted items are then tampered in one field.  It is the one valid input on which both equations END at the identity: the short form's accumulator, the long form's last
    addition and the half chains' join all meet their exceptional case on an expected-accept item.  The records come from the C oracle's signer (r = 0 is a status bit there,
    the outputs are still the reference's arithmetic).  Returns the planted indices."""
    if "sk" not in b:
        return np.zeros(0, dtype=np.int64)
    from tests import _oracle_c as OC
    rng = random.Random(seed ^ 0x5A5A)
    n = len(b["off"]) - 1
    idx = np.array(sorted(rng.sample(range(n), max(1, n // every))), dtype=np.int64)
    lo, hi = b["off"][idx].astype(np.int64), b["off"][idx + 1].astype(np.int64)
    msgs = [b["msgs"][a:z].tobytes() for a, z in zip(lo, hi)]
    mb, off = OC.pack_msgs(msgs)
    sg = OC.sign_batch(ver, mb, off, np.ascontiguousarray(b["sk"][idx]), np.zeros((len(idx), 32), dtype=np.uint8), nthreads=4)
    assert not sg["r_point"].any() and not sg["hashed_to_curve_r"].any() and (sg["status"] & 2).all()
    for k, i in enumerate(idx):
        v["msgs"][lo[k]:hi[k]] = b["msgs"][lo[k]:hi[k]]                 # (an earlier mutation may have flipped a message bit)
        for f in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r"):
            if f in v:
                v[f][i] = sg[f][k]
        t = rng.randrange(12) if tamper else 11
        if t == 0:
            v["s"][i, 31] ^= 1
        elif t == 1:
            v["c"][i, rng.randrange(32)] ^= 1 << rng.randrange(8)
        elif t == 2:
            v["nullifier"][i] = v["nullifier"][(i + 1) % n]
        elif t == 3 and "r_point" in v:
            v["r_point"][i] = v["pk"][i]
        elif t == 4 and "hashed_to_curve_r" in v:
            v["hashed_to_curve_r"][i] = v["nullifier"][i]
        elif t == 5 and hi[k] > lo[k]:
            v["msgs"][lo[k]] ^= 1
        # 6..11: the planted signature stays as it is (expected: accepted)
    return idx


def fuzz_non_zk_batch(ver, signed, b, seed):
    """verify_non_zk inputs from a signed batch: honest items and field-level mutations incl. what only the non-zk semantics allow
    (zero scalars, identity pk -> Err).  Expected results come from the C oracle."""
    rng = random.Random(seed)
    v = fuzz_verify_batch(ver, signed, b, seed)           # same mutation kinds on pk / nullifier / c / s / R / Hr / msg
    n = len(b["off"]) - 1
    for i in range(n):
        k = rng.randrange(24)
        if k == 0:                                         # s = 0, c = 0 with identity R, Hr: both equations hold, the hash decides
            v["s"][i] = 0; v["c"][i] = 0; v["r_point"][i] = 0; v["hashed_to_curve_r"][i] = 0
        elif k == 1:
            v["s"][i] = 0
        elif k == 2:
            v["c"][i] = 0
        elif k == 3:
            v["pk"][i] = 0                                 # Err(HashToCurveError)
    return v
