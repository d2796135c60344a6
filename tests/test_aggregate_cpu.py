"""The aggregate random-linear-combination check (include/plume_hip.h plume_aggregate_check) on the CPU: the product's per-lane
bodies (tests/devsim, bucket method) against the oracle's item-by-item DEFINITION -- the 72-byte record (flags, n_bad, the aggregate POINT) and
hash_ok must be identical, for honest batches (the point is the identity) and for corrupted ones (a specific non-identity point)."""
import hashlib

import numpy as np
import pytest

from tests import _devsim as DS
from tests import _fuzz
from tests import _oracle_c as OC
from tests import synth

SEED = hashlib.sha256(b"aggregate test seed").digest()


def _signed(n, ver, start=0):
    b = synth.sign_inputs(n, start=start)
    return b, OC.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])


def _both(ver, mode, v, seed=SEED, index_base=0, W=0):
    args = (ver, mode, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"], seed)
    want, want_ok = OC.aggregate_check(*args, index_base=index_base)
    got, got_ok = DS.aggregate_check(*args, index_base=index_base, W=W)
    assert np.array_equal(got_ok, want_ok)
    assert got.tobytes() == want.tobytes(), (OC.parse_aggregate_record(got), OC.parse_aggregate_record(want))
    return OC.parse_aggregate_record(want), want_ok


@pytest.mark.parametrize("ver,mode", [(1, 0), (1, 1), (2, 1)])
def test_honest_batch_sums_to_the_identity(ver, mode):
    b, sg = _signed(40, ver)
    v = dict(msgs=b["msgs"], off=b["off"], **{k: sg[k] for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")})
    r, hok = _both(ver, mode, v)
    assert r["all_ok"] == 1 and r["identity"] == 1 and r["n_bad"] == 0 and r["point"] == bytes(64) and hok.all()


@pytest.mark.parametrize("W", [4, 5, 8, 11, 16])      # the library uses 4, 8, 16; the bodies are correct for any width
def test_window_widths(W):
    b, sg = _signed(12, 1, start=100)
    v = dict(msgs=b["msgs"], off=b["off"], **{k: sg[k].copy() for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")})
    r, _ = _both(1, 0, v, W=W)
    assert r["all_ok"] == 1
    v["s"][5, 31] ^= 1                       # a false pair of equations, the hash still matches (s is not hashed)
    r, hok = _both(1, 0, v, W=W)
    assert r["all_ok"] == 0 and r["identity"] == 0 and r["n_bad"] == 0 and hok.all() and r["point"] != bytes(64)


@pytest.mark.parametrize("ver,mode,seed", [(1, 0, 11), (1, 1, 12), (2, 1, 13)])
def test_fuzzed_batches_match_the_definition(ver, mode, seed):
    b, sg = _signed(96, ver, start=1000 * seed)
    v = (_fuzz.fuzz_non_zk_batch if mode else _fuzz.fuzz_verify_batch)(ver, sg, b, seed)
    r, hok = _both(ver, mode, v, index_base=7 * seed)
    threaded, _ = OC.aggregate_check(ver, mode, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"], SEED, index_base=7 * seed, nthreads=5)
    assert OC.parse_aggregate_record(threaded) == r
    assert r["all_ok"] == 0 and 0 < r["n_bad"] < 96
    # the per-item verdict implies the hash verdict; an item that verifies has hash_ok
    if mode:
        ok = OC.verify_non_zk_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["s"], v["r_point"], v["hashed_to_curve_r"], v["c"])
    else:
        ok = OC.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"])
    assert np.all(hok[ok == 1] == 1)


def test_repeated_points_and_one_signer():
    """one signer, one message signed with different nonces, and literally repeated items: equal points meet inside a bucket (the checked additions'
    doubling branch); small windows make that certain"""
    b = synth.sign_inputs(24, start=50)
    for i in range(24):
        b["sk"][i] = b["sk"][0]
    for i in range(12, 24):                                  # items 12.. repeat items 0..11 entirely
        b["r"][i] = b["r"][i - 12]
    msgs = [b["msgs"][int(b["off"][i % 12]):int(b["off"][i % 12 + 1])].tobytes() for i in range(24)]
    buf, off = OC.pack_msgs(msgs)
    sg = OC.sign_batch(1, buf, off, b["sk"], b["r"])
    v = dict(msgs=buf, off=off, **{k: sg[k] for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")})
    for W in (4, 6):
        r, _ = _both(1, 0, v, W=W)
        assert r["all_ok"] == 1


def test_pieces_and_shards_add_up():
    """coefficients are indexed by the position in the whole batch: the record of a batch cut into pieces (carry) or shards (combine) equals the
    record of the batch"""
    b, sg = _signed(30, 1, start=300)
    v = dict(msgs=b["msgs"], off=b["off"], **{k: sg[k].copy() for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")})
    v["s"][3, 30] ^= 4; v["c"][20, 5] ^= 1; v["pk"][25] = 0x11      # false equation; hash mismatch; garbage point
    whole, _ = OC.aggregate_check(1, 0, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"], SEED)

    def piece(lo, hi, carry=None):
        off = (v["off"][lo:hi + 1] - v["off"][lo]).astype(np.uint64)
        msgs = np.ascontiguousarray(v["msgs"][int(v["off"][lo]):int(v["off"][hi]) + 1])
        rec, _ = DS.aggregate_check(1, 0, msgs, off, *(np.ascontiguousarray(v[k][lo:hi]) for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")), SEED,
                                    index_base=lo, carry=carry)
        return rec
    r1 = piece(0, 11)
    r2 = piece(11, 30, carry=r1)
    assert r2.tobytes() == whole.tobytes()
    shards = [piece(0, 7), piece(7, 19), piece(19, 30)]
    assert DS.aggregate_combine(shards).tobytes() == whole.tobytes()
    p = OC.parse_aggregate_record(whole)
    assert p["n_bad"] == 2 and p["all_ok"] == 0 and p["identity"] == 0


def test_empty_batch():
    z = np.zeros((0, 64), dtype=np.uint8); k = np.zeros((0, 32), dtype=np.uint8)
    rec, _ = DS.aggregate_check(1, 0, np.zeros(1, dtype=np.uint8), np.zeros(1, dtype=np.uint64), z, z, k, k, z, z, SEED)
    p = OC.parse_aggregate_record(rec)
    assert p["all_ok"] == 1 and p["identity"] == 1 and p["n_bad"] == 0
