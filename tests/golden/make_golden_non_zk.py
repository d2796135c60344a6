#!/usr/bin/env python3
"""Golden vectors for plume_arkworks' verify_non_zk (rust-arkworks/src/tests.rs:28-78), generated with the PYTHON oracle
(oracle/plume_oracle.py, itself pinned to the reference's KATs).  Output: tests/golden/golden_non_zk.json — inputs and the
expected result only (1 Ok(true), 0 Ok(false), 2 Err(HashToCurveError)).

    python tests/golden/make_golden_non_zk.py        (~1 min)

Items: the 64 + 64 seeded signatures of golden_batches.json (sign_v1 / sign_v2), each mutated by a kind chosen from its index:
honest, s ^ 1, digest_private ^ 1, nullifier of the previous item, r_point <-> hashed_to_curve_r, r_point negated (a V2 signature
whose hash still matches only if the given points are hashed), r_point of the previous item, pk = identity (Err), s >= n,
off-curve nullifier, all-zero scalars with identity points, digest_private = the other version's challenge.
"""
import json
import sys
from multiprocessing import Pool
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import plume_oracle as O  # noqa: E402

KINDS = ["honest", "s^1", "digest^1", "nullifier of the previous item", "r_point <-> hashed_to_curve_r", "r_point negated", "r_point of the previous item",
         "pk = identity", "s >= n", "nullifier off the curve", "zero scalars, identity points", "digest of the other version", "honest", "honest",
         "hashed_to_curve_r negated", "s = 0"]


def mutate(ver, i, it, prev, other):
    it = {k: it[k] for k in ("msg", "pk", "nullifier", "s", "r_point", "hashed_to_curve_r")} | {"digest_private": it["c"]}
    kind = KINDS[i % len(KINDS)]
    x = lambda h, bit=1: (int(h, 16) ^ bit).to_bytes(len(h) // 2, "big").hex()  # noqa: E731
    neg = lambda p: p[:64] + ((O.P - int(p[64:], 16)) % O.P).to_bytes(32, "big").hex()  # noqa: E731
    if kind == "s^1":
        it["s"] = x(it["s"])
    elif kind == "digest^1":
        it["digest_private"] = x(it["digest_private"])
    elif kind == "nullifier of the previous item":
        it["nullifier"] = prev["nullifier"]
    elif kind == "r_point <-> hashed_to_curve_r":
        it["r_point"], it["hashed_to_curve_r"] = it["hashed_to_curve_r"], it["r_point"]
    elif kind == "r_point negated":
        it["r_point"] = neg(it["r_point"])
    elif kind == "hashed_to_curve_r negated":
        it["hashed_to_curve_r"] = neg(it["hashed_to_curve_r"])
    elif kind == "r_point of the previous item":
        it["r_point"] = prev["r_point"]
    elif kind == "pk = identity":
        it["pk"] = "00" * 64
    elif kind == "s >= n":
        it["s"] = (O.N + (i % 3)).to_bytes(32, "big").hex()
    elif kind == "nullifier off the curve":
        it["nullifier"] = it["nullifier"][:64] + x(it["nullifier"][64:], 2)
    elif kind == "zero scalars, identity points":
        it["s"] = it["digest_private"] = "00" * 32
        it["r_point"] = it["hashed_to_curve_r"] = "00" * 64
    elif kind == "digest of the other version":
        it["digest_private"] = other["c"]
    elif kind == "s = 0":
        it["s"] = "00" * 32
    it["note"] = kind
    it["version"] = ver
    return it


def expect(it):
    P = O.pt_from_bytes
    b = bytes.fromhex
    pts = [P(b(it[k])) for k in ("pk", "nullifier", "r_point", "hashed_to_curve_r")]
    if any(p is not None and (p[0] >= O.P or p[1] >= O.P or not O.is_on_curve(p)) for p in pts):
        return dict(it, ok=0)                      # a point the reference's types cannot hold
    s, d = int(it["s"], 16), int(it["digest_private"], 16)
    if s >= O.N or d >= O.N:
        return dict(it, ok=0)                      # not an Fr element
    try:
        ok = int(O.verify_non_zk(it["version"], b(it["msg"]), pts[0], pts[1], s, pts[2], pts[3], d))
    except O.HashToCurveError:
        ok = 2
    return dict(it, ok=ok)


def main():
    gold = json.loads((ROOT / "tests" / "golden" / "golden_batches.json").read_text())
    items = []
    for ver in (1, 2):
        src, oth = gold[f"sign_v{ver}"], gold[f"sign_v{3 - ver}"]
        for i, it in enumerate(src):
            items.append(mutate(ver, i, it, src[i - 1], oth[i]))
    # The signature with nonce r = 0 (R = Hr = identity, s = c sk): both equations hold with every term ending at the identity and the hash is taken over .. || 00 || 00
    # -- verify_non_zk returns Ok(true) (rust-arkworks/src/tests.rs:28-78 has no nonce check either) -- followed by its single-field tamperings.
    for ver in (1, 2):
        for tag, skz, mz in (("", O.synth_sk(4242), b"nonce zero"), (", empty message", O.synth_sk(4243), b""), (", sk=1", 1, b"nonce zero")):
            sig = O.sign(ver, skz, 0, mz)
            assert sig["r_point"] is None and sig["hashed_to_curve_r"] is None
            z = dict(msg=mz.hex(), pk=O.pt_bytes(sig["pk"]).hex(), nullifier=O.pt_bytes(sig["nullifier"]).hex(), s=sig["s"].to_bytes(32, "big").hex(),
                     r_point="00" * 64, hashed_to_curve_r="00" * 64, digest_private=sig["c"].to_bytes(32, "big").hex(), note="r=0: R=Hr=identity (Ok(true))" + tag, version=ver)
            items.append(z)
            if tag:
                continue
            g_hex = O.pt_bytes(O.G).hex()
            flip = lambda h: h[:-2] + f"{int(h[-2:], 16) ^ 1:02x}"  # noqa: E731
            for note, kw in (("s ^ 1", dict(s=flip(z["s"]))), ("digest ^ 1", dict(digest_private=flip(z["digest_private"]))), ("r_point = G", dict(r_point=g_hex)),
                             ("hashed_to_curve_r = G", dict(hashed_to_curve_r=g_hex)), ("nullifier = G", dict(nullifier=g_hex)), ("pk = G", dict(pk=g_hex)),
                             ("r_point = pk", dict(r_point=z["pk"])), ("message extended", dict(msg=z["msg"] + "00"))):
                items.append(dict(z, **kw, note="r=0 signature, " + note))
    with Pool(8) as pool:
        out = pool.map(expect, items)
    assert all(it["ok"] == 1 for it in out if it["note"].startswith("r=0: ")) and all(it["ok"] == 0 for it in out if it["note"].startswith("r=0 signature, "))
    assert sum(1 for it in out if it["ok"] == 1) >= 20 and any(it["ok"] == 2 for it in out) and any(it["ok"] == 0 for it in out)
    (ROOT / "tests" / "golden" / "golden_non_zk.json").write_text(json.dumps(
        {"_generated_by": "tests/golden/make_golden_non_zk.py (python oracle, rust-arkworks/src/tests.rs:28-78)", "items": out}, indent=0))
    print(len(out), "items;", {k: sum(1 for it in out if it["ok"] == k) for k in (0, 1, 2)})


if __name__ == "__main__":
    main()
