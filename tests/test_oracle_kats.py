"""Pin the Python oracle against EVERY known-answer vector the reference's own tests hold for the hot path
(data in tests/golden/reference_kats.json, extracted by make_reference_kats.py)."""
from oracle import plume_oracle as O


def H(s):
    return int(s, 16)


def test_constants_match_reference(kats):
    c = kats["constants"]
    assert O.P == int(c["p_dec"]) and O.N == int(c["n_dec"])
    assert O.B == int(c["b"]) and O.GX == int(c["gx"]) and O.GY == int(c["gy"])
    assert O.ISO_A == int(c["iso_a"]) and O.ISO_B == int(c["iso_b"])
    assert O.Z == int(c["z"]) % O.P
    for name, tab in [("x_num", O.ISO_XNUM), ("x_den", O.ISO_XDEN), ("y_num", O.ISO_YNUM), ("y_den", O.ISO_YDEN)]:
        assert [int(v, 16) for v in c["iso_" + name]] == tab, name
    assert O.DST == c["dst"].encode()
    # the isogenous curve's generator is on E' (curves/mod.rs:74-77)
    gx, gy = int(c["iso_gx"]), int(c["iso_gy"])
    assert (gy * gy - gx**3 - O.ISO_A * gx - O.ISO_B) % O.P == 0
    assert O.is_on_curve(O.G)


def test_plume_fixed_vector_sign(kats):
    v = kats["plume_vector"]
    msg = v["msg_utf8"].encode()
    assert len(msg) == 29
    sk, r = H(v["sk"]), H(v["r"])
    for ver in (1, 2):
        sig = O.sign(ver, sk, r, msg)
        assert sig["status"] == 0
        assert sig["pk"] == (H(v["pk_x"]), H(v["pk_y"]))
        assert sig["r_point"] == (H(v["g_r_x"]), H(v["g_r_y"]))
        assert sig["h"] == (H(v["h_x"]), H(v["h_y"]))
        assert sig["hashed_to_curve_r"] == (H(v["h_r_x"]), H(v["h_r_y"]))
        assert sig["nullifier"] == (H(v["nullifier_x"]), H(v["nullifier_y"]))
        assert sig["c"] == H(v[f"c_v{ver}"]) and sig["s"] == H(v[f"s_v{ver}"])
        # arkworks-shaped call (pk supplied) gives the same values (rust-arkworks/src/tests.rs:266-299)
        sig2 = O.sign(ver, sk, r, msg, pk=sig["pk"])
        assert sig2 == sig
    assert v["verification_c_v1"] == v["c_v1"]


def test_plume_fixed_vector_verify(kats):
    """rust-k256/tests/verification.rs:25-107: verify() is true for V1 and V2 built from the signals."""
    v = kats["plume_vector"]
    msg = v["msg_utf8"].encode()
    sk, r = H(v["sk"]), H(v["r"])
    for ver in (1, 2):
        sig = O.sign(ver, sk, r, msg)
        args = dict(msg=msg, pk=sig["pk"], nul=sig["nullifier"], c=sig["c"], s=sig["s"])
        if ver == 1:
            args.update(r_point=sig["r_point"], hashed_to_curve_r=sig["hashed_to_curve_r"])
        assert O.verify(ver, **args)
        assert O.verify_non_zk(ver, msg, sig["pk"], sig["nullifier"], sig["s"], sig["r_point"], sig["hashed_to_curve_r"], sig["c"])
        # negatives (the reference has none; semantic sanity of the oracle itself)
        bad = dict(args); bad["s"] = (args["s"] ^ 1)
        assert not O.verify(ver, **bad)
        bad = dict(args); bad["c"] = (args["c"] ^ 1)
        assert not O.verify(ver, **bad)
        bad = dict(args); bad["msg"] = b"X" + msg[1:]
        assert not O.verify(ver, **bad)
        bad = dict(args); bad["nul"] = O.pt_mul(2, args["nul"])
        assert not O.verify(ver, **bad)
        bad = dict(args); bad["c"] = 0
        assert not O.verify(ver, **bad)
        bad = dict(args); bad["pk"] = (args["pk"][0], args["pk"][1] ^ 1)
        assert not O.verify(ver, **bad)


def test_h2c_literal_preimage(kats):
    k = kats["h2c_preimage"]
    pre = bytes.fromhex(k["preimage_hex"])
    assert len(pre) == 62
    assert O.hash_to_curve_bytes(pre) == (H(k["x"]), H(k["y"]))
    v = kats["plume_vector"]
    assert pre == v["msg_utf8"].encode() + O.sec1_compress((H(v["pk_x"]), H(v["pk_y"])))


def test_h2c_abc(kats):
    k = kats["h2c_abc"]
    assert O.hash_to_curve_bytes(b"abc") == (H(k["x"]), H(k["y"]))


def test_rfc9380_j81_empty(kats):
    k = kats["rfc9380_empty"]
    pt, (u0, u1, q0, q1) = O.hash_to_curve_bytes(b"", want_intermediates=True)
    assert (u0, u1) == (H(k["u0"]), H(k["u1"]))
    assert q0 == (H(k["q0_x"]), H(k["q0_y"])) and q1 == (H(k["q1_x"]), H(k["q1_y"]))
    assert pt == (H(k["p_x"]), H(k["p_y"])) == (int(k["p_x_dec"]), int(k["p_y_dec"]))
    assert u0 == int(k["u0_dec"])


def test_sec1_vectors(kats):
    assert O.sec1_compress(O.G).hex() == kats["enc_G"]["hex"]
    acc = None
    for kk, comp, uncomp in kats["sec1_kG"]["vectors"]:
        pt = O.pt_mul(kk, O.G)
        assert pt == acc, kk           # running sum cross-checks pt_mul against pt_add
        assert O.sec1_compress(pt).hex() == comp, kk
        if pt is None:
            assert uncomp == "00"
        else:
            assert uncomp == "04" + O.pt_bytes(pt).hex()
            assert O.sec1_decompress(bytes.fromhex(comp)) == pt
        acc = O.pt_add(acc, O.G)


def test_wasm_readme_vector(kats):
    w = kats["wasm_readme"]
    der = bytes.fromhex(w["sk_sec1_der"])
    sk = int.from_bytes(der[7:39], "big")          # SEC1 ECPrivateKey: 30 len 02 01 01 04 20 <32 bytes> ...
    v = kats["plume_vector"]
    assert sk == H(v["sk"])
    sig = O.sign(2, sk, H(v["r"]), v["msg_utf8"].encode())
    assert O.sec1_compress(sig["pk"]).hex() == w["pk_sec1"]
    assert O.sec1_compress(sig["nullifier"]).hex() == w["nullifier_sec1"]
    assert der[-65:] == b"\x04" + O.pt_bytes(sig["pk"])


def test_identity_semantics():
    """sk = 0 'signature' (pk = nullifier = identity) — the reference's verify accepts it if the hash matches
    (rust-k256/src/lib.rs:93-145 has no identity check); encodings shrink to the single byte 00."""
    msg = b"edge"
    r = 12345
    R = O.pt_mul(r, O.G)
    h = O.hash_to_curve(msg, None)
    hr = O.pt_mul(r, h)
    for ver in (1, 2):
        c = int.from_bytes(O.c_hash(ver, None, h, None, R, hr), "big") % O.N
        kw = dict(r_point=R, hashed_to_curve_r=hr) if ver == 1 else {}
        assert O.verify(ver, msg, None, None, c, r, **kw)
