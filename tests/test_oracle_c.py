"""Pin the plain-C oracle: reference KATs + the Python oracle's golden batches (bit-exact)."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

from tests import _oracle_c as OC

GOLD = json.loads((Path(__file__).parent / "golden" / "golden_batches.json").read_text())


def test_sha256():
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1000):
        d = bytes((i * 7 + n) & 0xFF for i in range(n))
        assert OC.sha256(d) == hashlib.sha256(d).digest()


def test_reference_kats(kats):
    v = kats["plume_vector"]
    msg = v["msg_utf8"].encode()
    mb, off = OC.pack_msgs([msg])
    sk = np.frombuffer(bytes.fromhex(v["sk"]), dtype=np.uint8).reshape(1, 32).copy()
    r = np.frombuffer(bytes.fromhex(v["r"]), dtype=np.uint8).reshape(1, 32).copy()
    for ver in (1, 2):
        o = OC.sign_batch(ver, mb, off, sk, r)
        assert o["status"][0] == 0
        assert o["pk"][0].tobytes().hex() == v["pk_x"] + v["pk_y"]
        assert o["r_point"][0].tobytes().hex() == v["g_r_x"] + v["g_r_y"]
        assert o["h"][0].tobytes().hex() == v["h_x"] + v["h_y"]
        assert o["hashed_to_curve_r"][0].tobytes().hex() == v["h_r_x"] + v["h_r_y"]
        assert o["nullifier"][0].tobytes().hex() == v["nullifier_x"] + v["nullifier_y"]
        assert o["c"][0].tobytes().hex() == v[f"c_v{ver}"] and o["s"][0].tobytes().hex() == v[f"s_v{ver}"]
        # arkworks-shaped: pk supplied
        o2 = OC.sign_batch(ver, mb, off, sk, r, pk_in=o["pk"])
        for k in o:
            assert np.array_equal(o[k], o2[k]), k
        ok = OC.verify_batch(ver, mb, off, o["pk"], o["nullifier"], o["c"], o["s"],
                             o["r_point"] if ver == 1 else None, o["hashed_to_curve_r"] if ver == 1 else None)
        assert ok[0] == 1
    k = kats["h2c_preimage"]
    assert OC.h2c_raw(bytes.fromhex(k["preimage_hex"]))["p"].hex() == k["x"] + k["y"]
    k = kats["h2c_abc"]
    assert OC.h2c_raw(b"abc")["p"].hex() == k["x"] + k["y"]
    k = kats["rfc9380_empty"]
    o = OC.h2c_raw(b"")
    assert o["u0"].hex() == k["u0"] and o["u1"].hex() == k["u1"]
    assert o["q0"].hex() == k["q0_x"] + k["q0_y"] and o["q1"].hex() == k["q1_x"] + k["q1_y"]
    assert o["p"].hex() == k["p_x"] + k["p_y"]
    g = bytes.fromhex(kats["sec1_kG"]["vectors"][1][2][2:])
    assert OC.sec1_compress(g).hex() == kats["enc_G"]["hex"]
    for kk, comp, uncomp in kats["sec1_kG"]["vectors"]:
        pt = OC.point_mul(kk.to_bytes(32, "big"), g)
        assert OC.sec1_compress(pt).hex() == comp
        assert (pt == bytes(64)) if kk == 0 else (pt.hex() == uncomp[2:])


@pytest.mark.parametrize("ver", [1, 2])
def test_golden_sign(ver):
    items = GOLD[f"sign_v{ver}"]
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    o = OC.sign_batch(ver, mb, off, OC.arr(items, "sk", 32), OC.arr(items, "r", 32), nthreads=4)
    for key, w in [("pk", 64), ("h", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]:
        assert np.array_equal(o[key], OC.arr(items, key, w)), key
    assert list(o["status"]) == [it["status"] for it in items]
    for it in items[:8]:
        raw = OC.h2c_raw(bytes.fromhex(it["msg"]) + OC.sec1_compress(bytes.fromhex(it["pk"])))
        assert (raw["u0"].hex(), raw["u1"].hex(), raw["q0"].hex(), raw["q1"].hex(), raw["p"].hex()) == (it["u0"], it["u1"], it["q0"], it["q1"], it["h"])


@pytest.mark.parametrize("ver", [1, 2])
def test_golden_verify(ver):
    items = GOLD[f"verify_v{ver}"]
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    ok = OC.verify_batch(ver, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32),
                         OC.arr(items, "r_point", 64) if ver == 1 else None, OC.arr(items, "hashed_to_curve_r", 64) if ver == 1 else None, nthreads=8)
    assert list(ok) == [it["ok"] for it in items]
    assert sum(ok) == 240


@pytest.mark.parametrize("ver", [1, 2])
def test_golden_edge(ver):
    items = [e for e in GOLD["edge"] if e["version"] == ver]
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    ok = OC.verify_batch(ver, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32),
                         OC.arr(items, "r_point", 64) if ver == 1 else None, OC.arr(items, "hashed_to_curve_r", 64) if ver == 1 else None, nthreads=8)
    bad = [(it["note"], int(o), it["ok"]) for it, o in zip(items, ok) if int(o) != it["ok"]]
    assert not bad, bad


NONZK = json.loads((Path(__file__).parent / "golden" / "golden_non_zk.json").read_text())["items"]


def non_zk_args(items):
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    return (mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "s", 32), OC.arr(items, "r_point", 64),
            OC.arr(items, "hashed_to_curve_r", 64), OC.arr(items, "digest_private", 32))


@pytest.mark.parametrize("ver", [1, 2])
def test_golden_verify_non_zk(ver):
    """plume_arkworks' verify_non_zk (rust-arkworks/src/tests.rs:28-78): C oracle == Python oracle's vectors, incl. Err (2)"""
    items = [it for it in NONZK if it["version"] == ver]
    ok = OC.verify_non_zk_batch(ver, *non_zk_args(items), nthreads=8)
    bad = [(it["note"], int(o), it["ok"]) for it, o in zip(items, ok) if int(o) != it["ok"]]
    assert not bad, bad
    assert {int(o) for o in ok} == {0, 1, 2}
