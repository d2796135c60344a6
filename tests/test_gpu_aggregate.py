"""GPU tests of the aggregate random-linear-combination pre-filter (include/plume_hip.h plume_aggregate_check): the 72-byte record
(flags, n_bad, the aggregate POINT) and hash_ok against the oracle's item-by-item definition at sizes the oracle does in seconds, and at BASELINE's
2^20 through linearity: honest items contribute the identity, so the batch's point must equal the sum of the oracle's single-item points of the
corrupted items."""
import hashlib

import numpy as np
import pytest

from oracle import plume_oracle as O
from tests import _fuzz, synth
from tests import _oracle_c as OC

pytestmark = pytest.mark.gpu
SEED = hashlib.sha256(b"gpu aggregate seed").digest()
KEYS = ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")


@pytest.fixture(scope="module")
def eng():
    import zk_nullifier_sig_amd as plume
    e = plume.Engine(0)
    yield e
    e.close()


def _args(v):
    return (v["msgs"], v["off"], *(v[k] for k in KEYS))


def _check(eng, ver, mode, v, seed=SEED):
    got = eng.aggregate_check(ver, *_args(v), seed=seed, mode=mode)
    want_rec, want_ok = OC.aggregate_check(ver, mode, *_args(v), seed, nthreads=16)
    want = OC.parse_aggregate_record(want_rec)
    assert np.array_equal(got["hash_ok"], want_ok), np.nonzero(got["hash_ok"] != want_ok)[0][:10]
    assert (int(got["all_ok"]), int(got["identity"]), got["n_bad"], got["point"]) == (want["all_ok"], want["identity"], want["n_bad"], want["point"])
    return got


@pytest.mark.parametrize("ver,mode", [(1, 0), (1, 1), (2, 1)])
def test_honest_and_fuzzed_vs_definition(eng, ver, mode):
    n = 1536 if mode else 33000            # 8-bit and 16-bit windows
    b = synth.sign_inputs(n, start=7_100_000 + 10_000 * ver)
    sg = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    v = dict(msgs=b["msgs"], off=b["off"], **{k: sg[k] for k in KEYS})
    r = _check(eng, ver, mode, v)
    assert r["all_ok"] and r["identity"] and r["n_bad"] == 0 and r["point"] == bytes(64)
    f = (_fuzz.fuzz_non_zk_batch if mode else _fuzz.fuzz_verify_batch)(ver, sg, b, seed=40 + ver + mode)
    r = _check(eng, ver, mode, f)
    assert not r["all_ok"] and 0 < r["n_bad"] < n
    # one flipped bit of one s: every hash still matches, only the aggregate equation can tell
    w = dict(v, s=sg["s"].copy())
    w["s"][n // 3, 31] ^= 1
    r = _check(eng, ver, mode, w)
    assert not r["all_ok"] and not r["identity"] and r["n_bad"] == 0 and r["hash_ok"].all()


def test_small_sizes_and_window_choice(eng):
    """n = 0, 1, 2, ... : the three window widths the library picks (4 bits below 64 items, 8 below 2^15, 16 from there)"""
    b = synth.sign_inputs(40000, start=7_200_000)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    for n in (0, 1, 2, 19, 63, 64, 65, 700, 32767, 32768, 40000):
        off = b["off"][:n + 1]
        v = dict(msgs=b["msgs"], off=off, **{k: sg[k][:n].copy() for k in KEYS})
        r = _check(eng, 1, 0, v)
        assert r["all_ok"]
        if n:
            v["s"][n - 1, 0] ^= 0x40
            r = _check(eng, 1, 0, v)
            assert not r["all_ok"] and not r["identity"]


def test_one_signer_many_messages(eng):
    """equal points meet inside buckets (one pk for the whole batch, repeated items): the checked additions' doubling branch"""
    n = 512
    b = synth.sign_inputs(n, start=7_300_000)
    b["sk"][:] = b["sk"][0]
    b["r"][n // 2:] = b["r"][:n // 2]
    b["msgs"][32 * (n // 2):32 * n] = b["msgs"][:32 * (n // 2)]
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    assert np.array_equal(sg["nullifier"][0], sg["nullifier"][n // 2])
    v = dict(msgs=b["msgs"], off=b["off"], **{k: sg[k] for k in KEYS})
    assert _check(eng, 1, 0, v)["all_ok"]


def _pt_add_records(recs):
    acc = None
    for r in recs:
        p = O.pt_from_bytes(OC.parse_aggregate_record(r)["point"])
        acc = O.pt_add(acc, p)
    return O.pt_bytes(acc)


def test_2p20_batch_by_linearity(eng):
    """BASELINE's size.  Honest batch: the identity.  Then 8 corrupted items among 2^20: the batch's aggregate point must be the sum of the oracle's
    single-item points of the 7 that take part (the 8th has a garbage pk: not representable, excluded, counted in n_bad)"""
    n = 1 << 20
    b = synth.sign_inputs(n, start=8_000_000)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = dict(msgs=b["msgs"], off=b["off"], **{k: sg[k].copy() for k in KEYS})
    r = eng.aggregate_check(1, *_args(v), seed=SEED)
    assert r["all_ok"] and r["identity"] and r["n_bad"] == 0 and r["hash_ok"].all()
    idx_s = [0, 4097, 333_333, 777_777, n - 1]
    idx_c = [65_536, 900_001]
    for i in idx_s:
        v["s"][i, 17] ^= 0x10
    for i in idx_c:
        v["c"][i, 3] ^= 0x01
    v["pk"][123_456] = 0x5A
    r = eng.aggregate_check(1, *_args(v), seed=SEED)
    assert not r["all_ok"] and not r["identity"] and r["n_bad"] == 3
    bad = np.nonzero(r["hash_ok"] == 0)[0]
    assert list(bad) == sorted(idx_c + [123_456])
    singles = []
    for i in idx_s + idx_c:
        rec, _ = OC.aggregate_check(1, 0, v["msgs"][32 * i:32 * i + 32].copy(), np.array([0, 32], dtype=np.uint64), *(v[k][i:i + 1].copy() for k in KEYS), SEED, index_base=i)
        singles.append(rec)
    assert r["point"] == _pt_add_records(singles)
    # the per-item verify agrees on who the culprits are
    ok = eng.verify_batch(1, *_args(v))
    assert list(np.nonzero(ok == 0)[0]) == sorted(idx_s + idx_c + [123_456])


def test_pieces_shards_and_device_form_give_the_same_record(eng):
    import torch
    import zk_nullifier_sig_amd as plume
    n = 50_000
    b = synth.sign_inputs(n, start=7_400_000)
    sg = eng.sign_batch(2, b["msgs"], b["off"], b["sk"], b["r"])
    v = dict(msgs=b["msgs"], off=b["off"], **{k: sg[k].copy() for k in KEYS})
    v["s"][31_000, 9] ^= 2; v["c"][7, 0] ^= 0x80; v["r_point"][49_999] = 0
    want = eng.aggregate_check(2, *_args(v), seed=SEED, mode=1)
    assert not want["all_ok"] and want["n_bad"] >= 1
    e2 = plume.Engine(0)
    m = plume.Engine([0, 0, 0])
    try:
        e2.set_host_first_piece(3000); e2.set_host_piece(9000)              # 3000, 9000, 9000, ... pieces carrying the running record
        for e in (e2, m):
            got = e.aggregate_check(2, *_args(v), seed=SEED, mode=1)
            assert (got["all_ok"], got["identity"], got["n_bad"], got["point"]) == (want["all_ok"], want["identity"], want["n_bad"], want["point"])
            assert np.array_equal(got["hash_ok"], want["hash_ok"])
    finally:
        e2.close(); m.close()
    # device-resident form, two calls with disjoint coefficient ranges: their points add up to the batch's
    dev = "cuda:0"
    t = {k: torch.from_numpy(v[k]).to(dev) for k in KEYS}
    msgs = torch.from_numpy(v["msgs"]).to(dev)
    recs = []
    for lo, hi in ((0, 20_000), (20_000, n)):
        off = torch.from_numpy((v["off"][lo:hi + 1] - v["off"][lo]).astype(np.int64)).to(dev)
        res = torch.zeros(72, dtype=torch.uint8, device=dev)
        hok = torch.zeros(hi - lo, dtype=torch.uint8, device=dev)
        eng.aggregate_check_device(2, 1, hi - lo, msgs[32 * lo:], off, 32 * (hi - lo), *(t[k][lo:hi].contiguous() for k in KEYS), SEED, lo, hok, res)
        torch.cuda.synchronize()
        recs.append(res.cpu().numpy())
        assert np.array_equal(hok.cpu().numpy(), want["hash_ok"][lo:hi])
    assert _pt_add_records(recs) == want["point"]
    assert sum(OC.parse_aggregate_record(r)["n_bad"] for r in recs) == want["n_bad"]


def test_argument_errors(eng):
    import zk_nullifier_sig_amd as plume
    b = synth.sign_inputs(4, start=7_500_000)
    sg = eng.sign_batch(2, b["msgs"], b["off"], b["sk"], b["r"])
    v = dict(msgs=b["msgs"], off=b["off"], **{k: sg[k] for k in KEYS})
    with pytest.raises(plume.PlumeHipError, match="GIVEN"):
        eng.aggregate_check(2, *_args(v), seed=SEED, mode=0)           # V2 verify recomputes R, Hr: nothing to aggregate over
    with pytest.raises(plume.PlumeHipError, match="mode"):
        eng.aggregate_check(1, *_args(v), seed=SEED, mode=3)
    with pytest.raises(ValueError):
        eng.aggregate_check(1, *_args(v), seed=b"short", mode=1)
    assert eng.aggregate_check(2, *_args(v), seed=SEED, mode=1)["all_ok"]
