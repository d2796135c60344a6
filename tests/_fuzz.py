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
    return v


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
