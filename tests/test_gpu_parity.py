"""GPU parity tests (run on the MI355X box: `pytest -m gpu`).  Everything goes through the C ABI of
libplume_hip.so (zk_nullifier_sig_amd.Engine is a ctypes binding of include/plume_hip.h) and is compared
bit-exactly with the oracles; nothing here reads /root/reference."""
import json
import random
from pathlib import Path

import numpy as np
import pytest

from oracle import plume_oracle as O
from tests import _oracle_c as OC
from tests import synth

pytestmark = pytest.mark.gpu

GOLD = json.loads((Path(__file__).parent / "golden" / "golden_batches.json").read_text())
N_ORDER = synth.N


@pytest.fixture(scope="module")
def eng():
    import zk_nullifier_sig_amd as plume
    e = plume.Engine(0)
    yield e
    e.close()


def _verify_args(items, ver):
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    return (ver, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32),
            OC.arr(items, "r_point", 64) if ver == 1 else None, OC.arr(items, "hashed_to_curve_r", 64) if ver == 1 else None)


def test_native_library_is_loaded(eng):
    """the product path is the HIP library, not a fallback"""
    import zk_nullifier_sig_amd as plume
    maps = Path("/proc/self/maps").read_text()
    assert str(plume.library_path()) in maps
    assert "gfx950" in eng.version()


# ------------------------------------------------------------------------------------------ reference KATs
def test_reference_fixed_vector_through_facade(eng, kats):
    """BASELINE config 1 / rust-k256/tests/signing.rs:48-64 and verification.rs:25-107, on the GPU"""
    import zk_nullifier_sig_amd as plume
    v = kats["plume_vector"]

    class Mock:  # rust-k256/tests/signing.rs:23-44
        def fill_bytes(self, n):
            assert n == 32
            return bytes.fromhex(v["r"])

    sk = plume.SecretKey.from_bytes(bytes.fromhex(v["sk"]))
    msg = v["msg_utf8"].encode()
    s1 = plume.PlumeSignature.sign_v1(sk, msg, Mock(), eng)
    s2 = plume.PlumeSignature.sign_v2(sk, msg, Mock(), eng)
    assert (s1.c.to_bytes().hex(), s1.s.to_bytes().hex()) == (v["c_v1"], v["s_v1"])
    assert (s2.c.to_bytes().hex(), s2.s.to_bytes().hex()) == (v["c_v2"], v["s_v2"])
    assert s1.v1specific is not None and s2.v1specific is None
    for sg in (s1, s2):
        assert sg.pk.to_bytes64().hex() == v["pk_x"] + v["pk_y"]
        assert sg.nullifier.to_bytes64().hex() == v["nullifier_x"] + v["nullifier_y"]
        assert sg.verify(eng)
    assert s1.v1specific.r_point.to_bytes64().hex() == v["g_r_x"] + v["g_r_y"]
    assert s1.v1specific.hashed_to_curve_r.to_bytes64().hex() == v["h_r_x"] + v["h_r_y"]
    # arkworks shape (rust-arkworks/src/tests.rs:266-299)
    for ver, c, s in ((plume.PlumeVersion.V1, v["c_v1"], v["s_v1"]), (plume.PlumeVersion.V2, v["c_v2"], v["s_v2"])):
        pub, prv = plume.sign_with_r((s1.pk, sk.value), msg, int(v["r"], 16), ver, eng)
        assert prv.digest_private == int(c, 16) and pub.s == int(s, 16)
        assert pub.nullifier == s1.nullifier and prv.r_point == s1.v1specific.r_point
    # tampered signature is rejected
    bad = plume.PlumeSignature(msg, s1.pk, s1.nullifier, s1.c, plume.NonZeroScalar(s1.s.value ^ 1), s1.v1specific)
    assert not bad.verify(eng)


def test_h2c_kats(eng, kats):
    v = kats["plume_vector"]
    mb, off = OC.pack_msgs([v["msg_utf8"].encode()])
    pk = np.frombuffer(bytes.fromhex(v["pk_x"] + v["pk_y"]), dtype=np.uint8).reshape(1, 64).copy()
    assert eng.hash_to_curve_batch(mb, off, pk)[0].tobytes().hex() == v["h_x"] + v["h_y"]
    raw = [b"", b"abc", bytes.fromhex(kats["h2c_preimage"]["preimage_hex"])]
    mb, off = OC.pack_msgs(raw)
    h = eng.hash_to_curve_batch(mb, off, None)
    assert h[0].tobytes().hex() == kats["rfc9380_empty"]["p_x"] + kats["rfc9380_empty"]["p_y"]
    assert h[1].tobytes().hex() == kats["h2c_abc"]["x"] + kats["h2c_abc"]["y"]
    assert h[2].tobytes().hex() == kats["h2c_preimage"]["x"] + kats["h2c_preimage"]["y"]


def test_sec1_kG_vectors_via_sign(eng, kats):
    """k*G for k = 1..99 (rust-arkworks/src/tests/test_vectors.rs) through the sign path's pk = sk*G"""
    vec = [v for v in kats["sec1_kG"]["vectors"] if v[0] > 0]
    n = len(vec)
    mb, off = OC.pack_msgs([b"m"] * n)
    sk = np.frombuffer(b"".join(k.to_bytes(32, "big") for k, _, _ in vec), dtype=np.uint8).reshape(n, 32).copy()
    o = eng.sign_batch(2, mb, off, sk, sk)
    for i, (k, comp, uncomp) in enumerate(vec):
        assert o["pk"][i].tobytes().hex() == uncomp[2:], k
        assert o["r_point"][i].tobytes().hex() == uncomp[2:], k


# ------------------------------------------------------------------------------------------ golden fixtures
@pytest.mark.parametrize("ver", [1, 2])
def test_golden_verify(eng, ver):
    items = GOLD[f"verify_v{ver}"]
    ok = eng.verify_batch(*_verify_args(items, ver))
    assert list(ok) == [it["ok"] for it in items]


@pytest.mark.parametrize("ver", [1, 2])
def test_golden_edge_cases(eng, ver):
    """ragged / empty messages, identity and off-curve points, non-canonical coordinates, zero and >= n scalars,
    sk = 0 forgery the reference accepts, sk = 1 (pk = G), r = sk, sk = n-1"""
    items = [e for e in GOLD["edge"] if e["version"] == ver]
    ok = eng.verify_batch(*_verify_args(items, ver))
    bad = [(it["note"], int(o), it["ok"]) for it, o in zip(items, ok) if int(o) != it["ok"]]
    assert not bad, bad


@pytest.mark.parametrize("ver", [1, 2])
def test_golden_sign(eng, ver):
    items = GOLD[f"sign_v{ver}"]
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    o = eng.sign_batch(ver, mb, off, OC.arr(items, "sk", 32), OC.arr(items, "r", 32))
    for key, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]:
        assert np.array_equal(o[key], OC.arr(items, key, w)), key
    assert list(o["status"]) == [it["status"] for it in items]
    h = eng.hash_to_curve_batch(mb, off, OC.arr(items, "pk", 64))
    assert np.array_equal(h, OC.arr(items, "h", 64))
    o2 = eng.sign_batch(ver, mb, off, OC.arr(items, "sk", 32), OC.arr(items, "r", 32), pk_in=OC.arr(items, "pk", 64))   # BASELINE config 5 shape
    for k in o:
        assert np.array_equal(o[k], o2[k]), k


# ------------------------------------------------------------------------------------------ seeded parity vs the C oracle
def test_sign_edge_status_vs_oracle(eng):
    rng = random.Random(9)
    sks = [0, N_ORDER, N_ORDER + 5, 1, N_ORDER - 1, 2**256 - 1, rng.randrange(1, N_ORDER), rng.randrange(1, N_ORDER)]
    rs = [rng.randrange(1, N_ORDER), rng.randrange(1, N_ORDER), 0, N_ORDER - 1, 1, 7, N_ORDER, rng.randrange(1, N_ORDER)]
    msgs = [bytes(rng.randrange(256) for _ in range(l)) for l in (0, 1, 32, 55, 56, 64, 100, 257)]
    mb, off = OC.pack_msgs(msgs)
    sk = np.frombuffer(b"".join(x.to_bytes(32, "big") for x in sks), dtype=np.uint8).reshape(-1, 32).copy()
    r = np.frombuffer(b"".join(x.to_bytes(32, "big") for x in rs), dtype=np.uint8).reshape(-1, 32).copy()
    for ver in (1, 2):
        a = eng.sign_batch(ver, mb, off, sk, r)
        b = OC.sign_batch(ver, mb, off, sk, r)
        for k in a:
            assert np.array_equal(a[k], b[k]), (ver, k)


@pytest.mark.parametrize("ver", [1, 2])
def test_ragged_batch_vs_oracle(eng, ver):
    """1000 items, message lengths 0..200, non-multiple-of-wave batch size"""
    rng = random.Random(100 + ver)
    n = 1000
    b = synth.sign_inputs(n, start=5000)
    msgs = [bytes(rng.randrange(256) for _ in range(rng.randrange(0, 201))) for _ in range(n)]
    mb, off = OC.pack_msgs(msgs)
    got = eng.sign_batch(ver, mb, off, b["sk"], b["r"])
    want = OC.sign_batch(ver, mb, off, b["sk"], b["r"], nthreads=16)
    for k in got:
        assert np.array_equal(got[k], want[k]), k
    bb = dict(msgs=mb, off=off)
    v = synth.corrupt_for_verify(ver, bb, got, start=5000)
    ok = eng.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"))
    want_ok = OC.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"), nthreads=16)
    assert np.array_equal(ok, want_ok)
    assert np.array_equal(ok, synth.expected_ok(n, 5000))


def test_config2_verify_v1_2p16_bit_exact(eng):
    """BASELINE config 2: batch 2^16 V1 verify, ok[] byte-identical to the CPU restatement over the whole batch"""
    n = 1 << 16
    b = synth.sign_inputs(n)
    signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, signed)
    ok = eng.verify_batch(1, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"])
    assert np.array_equal(ok, synth.expected_ok(n))
    import os
    want = OC.verify_batch(1, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"], nthreads=min(64, os.cpu_count() or 8))
    assert np.array_equal(ok, want)
    # and the first 256 are the committed Python-oracle fixture
    items = GOLD["verify_v1"]
    assert np.array_equal(v["pk"][:256], OC.arr(items, "pk", 64)) and np.array_equal(v["s"][:256], OC.arr(items, "s", 32))
    assert list(ok[:256]) == [it["ok"] for it in items]


# ------------------------------------------------------------------------------------------ full-size properties
@pytest.mark.parametrize("ver,logn", [(1, 20), (2, 20)])
def test_full_size_properties(eng, ver, logn):
    """BASELINE configs 3/4 sizes (2^20 per GPU): sign -> verify round trip, corruption pattern, idempotence,
    a checksum of checksums across chunkings, and a seeded sample against the oracle"""
    import hashlib
    n = 1 << logn
    b = synth.sign_inputs(n)
    signed = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    assert not signed["status"].any()
    v = synth.corrupt_for_verify(ver, b, signed)
    args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"))
    ok = eng.verify_batch(*args)
    assert np.array_equal(ok, synth.expected_ok(n))
    # idempotence + independence from the chunking of the batch
    digest = hashlib.sha256(b"".join(signed[k].tobytes() for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r"))).hexdigest()
    eng.set_chunk(300_000)
    try:
        signed2 = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
        ok2 = eng.verify_batch(*args)
    finally:
        eng.set_chunk(1 << 20)
    assert hashlib.sha256(b"".join(signed2[k].tobytes() for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r"))).hexdigest() == digest
    assert np.array_equal(ok, ok2)
    # seeded sample vs the oracle (sign outputs and verify bits)
    rng = np.random.default_rng(7 + ver)
    idx = np.sort(rng.choice(n, size=2048, replace=False))
    sub_msgs = np.concatenate([b["msgs"][32 * i:32 * i + 32] for i in idx] + [np.zeros(16, np.uint8)])
    sub_off = np.arange(len(idx) + 1, dtype=np.uint64) * 32
    want = OC.sign_batch(ver, sub_msgs, sub_off, b["sk"][idx], b["r"][idx], nthreads=16)
    for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r", "status"):
        assert np.array_equal(signed[k][idx], want[k]), k
    sub_vmsgs = np.concatenate([v["msgs"][32 * i:32 * i + 32] for i in idx] + [np.zeros(16, np.uint8)])
    want_ok = OC.verify_batch(ver, sub_vmsgs, sub_off, v["pk"][idx], v["nullifier"][idx], v["c"][idx], v["s"][idx],
                              v["r_point"][idx] if ver == 1 else None, v["hashed_to_curve_r"][idx] if ver == 1 else None, nthreads=16)
    assert np.array_equal(ok[idx], want_ok)


def test_device_resident_api_matches_host_api(eng):
    """the *_device entry points (inputs already in HBM, caller's stream) give the same bytes as the host-pointer ones"""
    import torch
    n = 5000
    b = synth.sign_inputs(n, start=777)
    host = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    msgs, off, sk, r = t(b["msgs"]), t(b["off"].view(np.int64)), t(b["sk"]), t(b["r"])
    out = {k: torch.zeros((n, w), dtype=torch.uint8, device=dev) for k, w in
           [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
    status = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng.sign_batch_device(1, n, msgs, off, 32 * n, sk, r, None, out["pk"], out["nullifier"], out["c"], out["s"], out["r_point"], out["hashed_to_curve_r"], status)
    torch.cuda.synchronize()
    for k in out:
        assert np.array_equal(out[k].cpu().numpy(), host[k]), k
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng.set_stage_timing(True)                               # off by default (library 0.5)
    eng.verify_batch_device(1, n, msgs, off, 32 * n, out["pk"], out["nullifier"], out["c"], out["s"], out["r_point"], out["hashed_to_curve_r"], ok)
    torch.cuda.synchronize()
    assert bool(ok.all())
    st = eng.last_stage_times()
    eng.set_stage_timing(False)
    # (a call of at most 2^16 items runs its scalar stage inside the two-role ingest kernel: one stage, one launch less)
    assert [s for s, _ in st] == (["verify_ingest_h2c+scalars"] if n <= 65536 else ["verify_ingest_h2c", "verify_scalars"]) + ["tables", "verify_msm", "verify_finalize"] and all(ms > 0 for _, ms in st)


def test_empty_batch(eng):
    mb, off = OC.pack_msgs([])
    z = lambda w: np.zeros((0, w), dtype=np.uint8)  # noqa: E731
    assert len(eng.verify_batch(2, mb, off, z(64), z(64), z(32), z(32))) == 0
    assert len(eng.sign_batch(1, mb, off, z(32), z(32))["status"]) == 0


@pytest.mark.parametrize("ver", [1, 2])
def test_sec1_compressed_ingest(eng, ver):
    """33-byte SEC1 records decompressed + validated on the GPU (plume_verify_batch_sec1)"""
    from tests import _sec1
    items = GOLD[f"verify_v{ver}"] + [e for e in GOLD["edge"] if e["version"] == ver and "off curve" not in e["note"] and "non-canonical" not in e["note"] and "= p" not in e["note"]]
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    c33 = lambda k: _sec1.compress(OC.arr(items, k, 64))  # noqa: E731
    ok = eng.verify_batch_sec1(ver, mb, off, c33("pk"), c33("nullifier"), OC.arr(items, "c", 32), OC.arr(items, "s", 32),
                               c33("r_point") if ver == 1 else None, c33("hashed_to_curve_r") if ver == 1 else None)
    assert list(ok) == [it["ok"] for it in items]
    cases = _sec1.malformed_cases(ver, GOLD[f"verify_v{ver}"])
    mb, off = OC.pack_msgs([c[0] for c in cases])
    col = lambda j, w: np.frombuffer(b"".join(c[j] for c in cases), dtype=np.uint8).reshape(-1, w).copy()  # noqa: E731
    ok = eng.verify_batch_sec1(ver, mb, off, col(1, 33), col(2, 33), col(3, 32), col(4, 32), col(5, 33) if ver == 1 else None, col(6, 33) if ver == 1 else None)
    want = [int(_sec1.oracle_verify_sec1(ver, *c[:7])) for c in cases]
    bad = [(c[7], int(o), w) for c, o, w in zip(cases, ok, want) if int(o) != w]
    assert not bad, bad
    # a larger seeded batch: compressed ingest == 64-byte ingest
    n = 20000
    b = synth.sign_inputs(n, start=123456)
    signed = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(ver, b, signed, start=123456)
    a = eng.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"))
    bb = eng.verify_batch_sec1(ver, v["msgs"], v["off"], _sec1.compress(v["pk"]), _sec1.compress(v["nullifier"]), v["c"], v["s"],
                               _sec1.compress(v["r_point"]) if ver == 1 else None, _sec1.compress(v["hashed_to_curve_r"]) if ver == 1 else None)
    assert np.array_equal(a, bb) and np.array_equal(a, synth.expected_ok(n, 123456))


@pytest.mark.parametrize("ver", [1, 2])
def test_fuzzed_verify_batch_vs_oracle(eng, ver):
    """8192 items: honest signatures with random mutations (bit flips, garbage records, identities, boundary scalars,
    negated / non-canonical points, cross-item fields, swapped fields) — ok[] must equal the C oracle's byte for byte"""
    from tests import _fuzz
    import os
    n = 8192
    b = synth.sign_inputs(n, start=900000)
    rng = random.Random(4242 + ver)
    msgs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 31, 32, 33, 55, 56, 64, 100]))) for _ in range(n)]
    mb, off = OC.pack_msgs(msgs)
    signed = eng.sign_batch(ver, mb, off, b["sk"], b["r"])
    v = _fuzz.fuzz_verify_batch(ver, signed, dict(msgs=mb, off=off), seed=99 + ver)
    args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"] if ver == 1 else None, v["hashed_to_curve_r"] if ver == 1 else None)
    got = eng.verify_batch(*args)
    want = OC.verify_batch(*args, nthreads=min(64, os.cpu_count() or 8))
    diff = np.nonzero(got != want)[0]
    assert len(diff) == 0, diff[:10]
    assert 0.25 * n < int(got.sum()) < 0.75 * n     # the batch really mixes accepted and rejected items


def test_odd_batch_sizes_and_long_messages(eng):
    """batch sizes around wavefront / workgroup boundaries, and messages far longer than one SHA block"""
    rng = random.Random(2718)
    for n in (1, 2, 63, 64, 65, 127, 255, 256, 257, 511, 513):
        b = synth.sign_inputs(n, start=31337)
        got = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
        want = OC.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"], nthreads=8)
        for k in got:
            assert np.array_equal(got[k], want[k]), (n, k)
        ok = eng.verify_batch(1, b["msgs"], b["off"], got["pk"], got["nullifier"], got["c"], got["s"], got["r_point"], got["hashed_to_curve_r"])
        assert ok.all() and len(ok) == n
    lens = [0, 1, 1000, 4096, 65536, 100001, 55, 56]
    msgs = [bytes(rng.randrange(256) for _ in range(l)) for l in lens]
    mb, off = OC.pack_msgs(msgs)
    b = synth.sign_inputs(len(lens), start=99)
    for ver in (1, 2):
        got = eng.sign_batch(ver, mb, off, b["sk"], b["r"])
        want = OC.sign_batch(ver, mb, off, b["sk"], b["r"], nthreads=8)
        for k in got:
            assert np.array_equal(got[k], want[k]), (ver, k)
        ok = eng.verify_batch(ver, mb, off, got["pk"], got["nullifier"], got["c"], got["s"], got["r_point"] if ver == 1 else None, got["hashed_to_curve_r"] if ver == 1 else None)
        assert ok.all()


@pytest.mark.parametrize("ver", [1, 2])
def test_host_pieces_do_not_change_results(eng, ver):
    """host-pointer calls are pipelined in pieces (plume_set_host_piece): 3001 ragged items through 1 / 5 / 47 pieces, and 40 items
    through pieces of ONE item, give the C oracle's bytes every time, for sign, verify, SEC1 verify and hash_to_curve"""
    from tests import _sec1
    from zk_nullifier_sig_amd import capi
    rng = random.Random(4242 + ver)
    n = 3001
    b = synth.sign_inputs(n, start=77000)
    msgs = [bytes(rng.randrange(256) for _ in range(rng.randrange(1, 90))) for _ in range(n)]
    mb, off = OC.pack_msgs(msgs)
    want = OC.sign_batch(ver, mb, off, b["sk"], b["r"], nthreads=16)
    v = synth.corrupt_for_verify(ver, dict(msgs=mb, off=off), want, start=77000)
    want_ok = synth.expected_ok(n, 77000)
    want_h = OC.hash_to_curve_batch(mb, off, want["pk"], nthreads=16)
    pts = ("pk", "nullifier", "r_point", "hashed_to_curve_r") if ver == 1 else ("pk", "nullifier")
    comp = {k: _sec1.compress(v[k]) for k in pts}

    def rows(a, w, cnt):
        return np.ascontiguousarray(np.asarray(a).reshape(-1, w)[:cnt])

    try:
        for piece, cnt in ((1 << 20, n), (700, n), (64, n), (1, 40)):
            m2, o2 = OC.pack_msgs(msgs[:cnt])
            eng.set_host_piece(piece)
            got = eng.sign_batch(ver, m2, o2, rows(b["sk"], 32, cnt), rows(b["r"], 32, cnt))
            for k in got:
                w = np.asarray(want[k])
                assert np.array_equal(np.asarray(got[k]).reshape(cnt, -1), w.reshape(n, -1)[:cnt]), (piece, k)
            vm, vo = OC.pack_msgs([bytes(v["msgs"][int(v["off"][i]):int(v["off"][i + 1])]) for i in range(cnt)])
            ok = eng.verify_batch(ver, vm, vo, rows(v["pk"], 64, cnt), rows(v["nullifier"], 64, cnt), rows(v["c"], 32, cnt), rows(v["s"], 32, cnt),
                                  rows(v["r_point"], 64, cnt) if ver == 1 else None, rows(v["hashed_to_curve_r"], 64, cnt) if ver == 1 else None)
            assert np.array_equal(ok, want_ok[:cnt]), piece
            ok = eng.verify_batch_sec1(ver, vm, vo, rows(comp["pk"], 33, cnt), rows(comp["nullifier"], 33, cnt), rows(v["c"], 32, cnt), rows(v["s"], 32, cnt),
                                       rows(comp["r_point"], 33, cnt) if ver == 1 else None, rows(comp["hashed_to_curve_r"], 33, cnt) if ver == 1 else None)
            assert np.array_equal(ok, want_ok[:cnt]), piece
            h = eng.hash_to_curve_batch(m2, o2, rows(want["pk"], 64, cnt))
            assert np.array_equal(np.asarray(h).reshape(cnt, 64), want_h[:cnt]), piece
    finally:
        eng.set_host_piece(capi.DEFAULT_HOST_PIECE)          # the shipped default (plume_capi.hip), so that later tests of this module run what ships


@pytest.mark.parametrize("ver", [1, 2])
def test_small_secret_keys_exceptional_additions(eng, ver):
    """pk = +-G, +-2G, ... : s*G - c*pk and s*H - c*nullifier meet p == +-q inside the multi-scalar chain every few dozen items.
    The hot loop's additions are unchecked; such a lane is detected at the end (Z = 0 mod p) and redone with checked additions
    (plume_ec.h jac_madd / msm_run).  8192 such items: signatures and verdicts byte-identical to the C oracle."""
    n = 8192
    b = synth.sign_inputs(n, start=31000)
    small = [1, 2, 3, 4, 7, 8, 9, 16, 17, 128, 129, 255, 256, 257, O.N - 1, O.N - 2, O.N - 8, O.N - 16, O.N - 128, O.N - 256]
    sk = np.stack([np.frombuffer(small[i % len(small)].to_bytes(32, "big"), dtype=np.uint8) for i in range(n)])
    want = OC.sign_batch(ver, b["msgs"], b["off"], sk, b["r"], nthreads=16)
    got = eng.sign_batch(ver, b["msgs"], b["off"], sk, b["r"])
    for k in got:
        assert np.array_equal(np.asarray(got[k]).reshape(n, -1), np.asarray(want[k]).reshape(n, -1)), k
    v = synth.corrupt_for_verify(ver, b, want, start=31000)
    ok = eng.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"))
    want_ok = OC.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v.get("r_point"), v.get("hashed_to_curve_r"), nthreads=16)
    assert np.array_equal(ok, want_ok)


@pytest.mark.parametrize("ver", [1, 2])
def test_sign_with_sec1_outputs(eng, ver):
    """plume_sign_batch_sec1: the same signatures with the points as 33-byte SEC1 records (incl. the identity public key of an out-of-range
    secret key), and they verify through the SEC1 ingest"""
    from tests import _sec1
    n = 777
    b = synth.sign_inputs(n, start=123456)
    b["sk"][5] = 0                                   # sk = 0: pk and the nullifier are the identity -> records 00 || zeros
    want = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    got = eng.sign_batch_sec1(ver, b["msgs"], b["off"], b["sk"], b["r"])
    for k in ("pk", "nullifier", "r_point", "hashed_to_curve_r"):
        assert np.array_equal(got[k], _sec1.compress(want[k])), k
    for k in ("c", "s", "status"):
        assert np.array_equal(got[k], want[k]), k
    assert bytes(got["pk"][5]) == bytes(33)
    ok = eng.verify_batch_sec1(ver, b["msgs"], b["off"], got["pk"], got["nullifier"], got["c"], got["s"],
                               got["r_point"] if ver == 1 else None, got["hashed_to_curve_r"] if ver == 1 else None)
    oracle_ok = OC.verify_batch(ver, b["msgs"], b["off"], want["pk"], want["nullifier"], want["c"], want["s"],
                                want["r_point"] if ver == 1 else None, want["hashed_to_curve_r"] if ver == 1 else None, nthreads=8)
    assert np.array_equal(ok, oracle_ok)
