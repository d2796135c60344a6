"""helpers for the SEC1-compressed ingest tests: 64-byte affine records -> 33-byte records, plus malformed encodings"""
import numpy as np

from oracle import plume_oracle as O


def compress(a64):
    a64 = np.ascontiguousarray(a64, dtype=np.uint8).reshape(-1, 64)
    out = np.zeros((len(a64), 33), dtype=np.uint8)
    ident = ~a64.any(axis=1)
    out[:, 0] = 2 + (a64[:, 63] & 1)
    out[:, 1:] = a64[:, :32]
    out[ident] = 0
    return out


def oracle_verify_sec1(ver, msg, pk33, nul33, c, s, r33=None, hr33=None):
    """reference semantics: a record that does not deserialize => no signature object => False"""
    def dec(b):
        b = bytes(b)
        if b[0] == 0:
            return None
        if b[0] not in (2, 3):
            raise ValueError
        x = int.from_bytes(b[1:], "big")
        if x >= O.P:
            raise ValueError
        y = pow((x**3 + 7) % O.P, (O.P + 1) // 4, O.P)
        if (y * y - x**3 - 7) % O.P:
            raise ValueError
        return (x, y if (y & 1) == (b[0] & 1) else O.P - y)
    try:
        pk, nul = dec(pk33), dec(nul33)
        kw = dict(r_point=dec(r33), hashed_to_curve_r=dec(hr33)) if ver == 1 else {}
    except ValueError:
        return False
    return O.verify(ver, bytes(msg), pk, nul, int.from_bytes(bytes(c), "big"), int.from_bytes(bytes(s), "big"), **kw)


def malformed_cases(ver, items):
    """(msg, pk33, nul33, c, s, r33, hr33, note) derived from honest golden items"""
    out = []
    h = lambda it, k: bytes.fromhex(it[k])  # noqa: E731
    base = [it for it in items if it["ok"]][:12]
    x_not_on_curve = next(x for x in range(1, 100) if pow((x**3 + 7) % O.P, (O.P - 1) // 2, O.P) != 1)
    muts = [
        ("honest", lambda r: r),
        ("tag 04", lambda r: bytes([4]) + r[1:]),
        ("tag 01", lambda r: bytes([1]) + r[1:]),
        ("tag 05", lambda r: bytes([5]) + r[1:]),
        ("tag ff", lambda r: bytes([0xFF]) + r[1:]),
        ("parity flipped (valid point, wrong one)", lambda r: bytes([r[0] ^ 1]) + r[1:]),
        ("x = p (non-canonical)", lambda r: bytes([r[0]]) + O.P.to_bytes(32, "big")),
        ("x >= p", lambda r: bytes([r[0]]) + (2**256 - 1).to_bytes(32, "big")),
        ("x with no point", lambda r: bytes([2]) + x_not_on_curve.to_bytes(32, "big")),
        ("identity (00 + junk)", lambda r: bytes([0]) + r[1:]),
        ("identity (00 + zeros)", lambda r: bytes(33)),
        ("x = 0 (no point: 7 is a non-residue?)", lambda r: bytes([2]) + bytes(32)),
    ]
    fields = ["pk", "nullifier"] + (["r_point", "hashed_to_curve_r"] if ver == 1 else [])
    for bi, it in enumerate(base):
        rec = {k: bytes(compress(np.frombuffer(h(it, k), dtype=np.uint8))[0]) for k in ["pk", "nullifier"] + (["r_point", "hashed_to_curve_r"] if ver == 1 else [])}
        name, fn = muts[bi % len(muts)]
        for f in fields:
            r = dict(rec)
            r[f] = fn(rec[f])
            out.append((h(it, "msg"), r["pk"], r["nullifier"], h(it, "c"), h(it, "s"), r.get("r_point"), r.get("hashed_to_curve_r"), f"{f}: {name}"))
    return out
