"""The optimised CPU leg of bench.py's cpu_baseline (oracle/plume_cpu_fast.c: GLV + wNAF + lazy 4x64 field + one inversion per signature) must give
exactly the plain oracle's verdicts: reference-pinned goldens, the edge cases, fuzzed batches, and its field / GLV primitives against Python."""
import json
import random
from pathlib import Path

import numpy as np
import pytest

from oracle import plume_oracle as O
from tests import _cpu_fast as CF
from tests import _fuzz, synth
from tests import _oracle_c as OC

GOLD = json.loads((Path(__file__).parent / "golden" / "golden_batches.json").read_text())
P, N = O.P, O.N
LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72


def test_field_and_glv_primitives():
    rng = random.Random(8)
    vals = [0, 1, 2, P - 1, P, P + 1, 2**256 - 1, 2**255, 2**32 + 977, 2**256 - 2**32 - 978] + [rng.randrange(2**256) for _ in range(200)]
    for a in vals:
        b = rng.choice(vals)
        assert CF.ff_op(0, a, b) == a * b % P and CF.ff_op(1, a) == a * a % P
        assert CF.ff_op(3, a, b) == (a + b) % P and CF.ff_op(4, a, b) == (a - b) % P
        if a % P:
            assert CF.ff_op(2, a) == pow(a, -1, P)
    for k in [1, 2, N - 1, N - 2, 2**128, 2**255, LAMBDA, (N - LAMBDA) % N] + [rng.randrange(1, N) for _ in range(300)]:
        m1, n1, m2, n2 = CF.glv(k)
        k1, k2 = (-m1 if n1 else m1), (-m2 if n2 else m2)
        assert (k1 + k2 * LAMBDA - k) % N == 0 and m1 < 2**129 and m2 < 2**129


def _args(items, ver):
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    return (ver, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32),
            OC.arr(items, "r_point", 64) if ver == 1 else None, OC.arr(items, "hashed_to_curve_r", 64) if ver == 1 else None)


@pytest.mark.parametrize("ver", [1, 2])
def test_goldens_and_edge_cases(ver):
    items = GOLD[f"verify_v{ver}"]
    assert list(CF.verify_batch(*_args(items, ver), nthreads=4)) == [it["ok"] for it in items]
    edge = [e for e in GOLD["edge"] if e["version"] == ver]
    got = CF.verify_batch(*_args(edge, ver), nthreads=2)
    bad = [(it["note"], int(o), it["ok"]) for it, o in zip(edge, got) if int(o) != it["ok"]]
    assert not bad, bad


@pytest.mark.parametrize("ver", [1, 2])
def test_fuzz_and_small_keys_vs_plain_oracle(ver):
    n = 2048
    b = synth.sign_inputs(n, start=1234567)
    small = [1, 2, 3, 4, 7, 8, 9, 16, 17, 128, 129, 255, 256, 257, N - 1, N - 2, N - 8, N - 16, N - 128]        # pk = +-G, +-2G, ...: exceptional additions
    for i in range(0, n, 16):
        b["sk"][i] = np.frombuffer(small[(i // 16) % len(small)].to_bytes(32, "big"), dtype=np.uint8)
    rng = random.Random(ver)
    msgs = [rng.randbytes(rng.choice([0, 1, 31, 32, 33, 64, 100])) for _ in range(n)]
    mb, off = OC.pack_msgs(msgs)
    signed = OC.sign_batch(ver, mb, off, b["sk"], b["r"], nthreads=8)
    v = _fuzz.fuzz_verify_batch(ver, signed, dict(msgs=mb, off=off), seed=40 + ver)
    args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"] if ver == 1 else None, v["hashed_to_curve_r"] if ver == 1 else None)
    got, want = CF.verify_batch(*args, nthreads=8), OC.verify_batch(*args, nthreads=8)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
    assert 0.2 * n < int(got.sum()) < 0.8 * n


def _salt_sign_inputs(b, n):
    """edge scalars the reference's types cannot hold (status bits) and tiny / huge keys (exceptional additions) salted into a sign batch"""
    vals = [0, 1, 2, 3, 7, 8, 9, 16, 17, 255, 256, N - 1, N - 2, N - 8, N, N + 1, 2**256 - 1, 2**255]
    for k, i in enumerate(range(3, n, 37)):
        b["sk" if k % 3 else "r"][i] = np.frombuffer(vals[k % len(vals)].to_bytes(32, "big"), dtype=np.uint8)
    for i in range(5, n, 101):
        b["r"][i] = b["sk"][i]                          # r = sk: R = pk, Hr = nullifier


@pytest.mark.parametrize("ver", [1, 2])
def test_fast_signer_vs_plain_oracle(ver):
    """the signer of the optimised CPU leg (the checker of the GPU's whole-batch sign parity at 2^20) gives the plain oracle's bytes: goldens, ragged messages,
    edge scalars, and the arkworks shape (pk supplied, incl. supplied garbage / identity / foreign keys)"""
    items = GOLD[f"sign_v{ver}"]
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    got = CF.sign_batch(ver, mb, off, OC.arr(items, "sk", 32), OC.arr(items, "r", 32), nthreads=4)
    for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]:
        assert np.array_equal(got[k], OC.arr(items, k, w)), k
    n = 1536
    b = synth.sign_inputs(n, start=777_000)
    _salt_sign_inputs(b, n)
    rng = random.Random(10 + ver)
    mb, off = OC.pack_msgs([rng.randbytes(rng.choice([0, 1, 31, 32, 33, 64, 100, 300])) for _ in range(n)])
    want = OC.sign_batch(ver, mb, off, b["sk"], b["r"], nthreads=8)
    got = CF.sign_batch(ver, mb, off, b["sk"], b["r"], nthreads=8)
    for k in got:
        assert np.array_equal(got[k], want[k]), (k, np.nonzero((got[k] != want[k]).reshape(n, -1).any(axis=1))[0][:8])
    assert (want["status"] != 0).sum() > 5
    # arkworks shape: pk supplied
    pk_in = want["pk"].copy()
    pk_in[7::64] = 0                                    # identity
    pk_in[9::64, 5] ^= 1                                # off the curve
    pk_in[11::64] = np.roll(want["pk"], 1, axis=0)[11::64]   # somebody else's key
    want = OC.sign_batch(ver, mb, off, b["sk"], b["r"], pk_in=pk_in, nthreads=8)
    got = CF.sign_batch(ver, mb, off, b["sk"], b["r"], pk_in=pk_in, nthreads=8)
    for k in got:
        assert np.array_equal(got[k], want[k]), (k, np.nonzero((got[k] != want[k]).reshape(n, -1).any(axis=1))[0][:8])


@pytest.mark.parametrize("ver", [1, 2])
def test_fast_verify_non_zk_vs_plain_oracle(ver):
    gold = json.loads((Path(__file__).parent / "golden" / "golden_non_zk.json").read_text())
    items = [it for it in gold["items"] if it["version"] == ver] if isinstance(gold, dict) and "items" in gold else None
    n = 2048
    b = synth.sign_inputs(n, start=888_000)
    rng = random.Random(20 + ver)
    mb, off = OC.pack_msgs([rng.randbytes(rng.choice([0, 1, 32, 33, 100])) for _ in range(n)])
    signed = OC.sign_batch(ver, mb, off, b["sk"], b["r"], nthreads=8)
    v = _fuzz.fuzz_non_zk_batch(ver, signed, dict(msgs=mb, off=off), seed=60 + ver)
    args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["s"], v["r_point"], v["hashed_to_curve_r"], v["c"])
    got, want = CF.verify_non_zk_batch(*args, nthreads=8), OC.verify_non_zk_batch(*args, nthreads=8)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
    assert (want == 1).sum() > 0.2 * n and (want == 0).sum() > 0.2 * n and (want == 2).sum() > 0
    if items:
        mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
        a = (ver, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "s", 32), OC.arr(items, "r_point", 64), OC.arr(items, "hashed_to_curve_r", 64),
             OC.arr(items, "digest_private", 32))
        assert list(CF.verify_non_zk_batch(*a, nthreads=2)) == [it["ok"] for it in items]


def test_fast_sec1_decompression_vs_python():
    from tests import _sec1
    rng = random.Random(5)
    pts = [O.pt_mul(rng.randrange(1, N), O.G) for _ in range(40)]
    recs, want = [], []
    for x, y in pts:
        recs.append(bytes([2 + (y & 1)]) + x.to_bytes(32, "big")); want.append((x, y))
        recs.append(bytes([3 - (y & 1)]) + x.to_bytes(32, "big")); want.append((x, P - y))
    recs += [bytes(33), bytes([0]) + rng.randbytes(32), bytes([4]) + pts[0][0].to_bytes(32, "big"), bytes([1]) + pts[0][0].to_bytes(32, "big"),
             bytes([2]) + P.to_bytes(32, "big"), bytes([3]) + (2**256 - 1).to_bytes(32, "big"), bytes([2]) + bytes(32)]
    want += [None, None, "bad", "bad", "bad", "bad", "bad"]
    for x in range(1, 30):
        sq = pow((x**3 + 7) % P, (P - 1) // 2, P) == 1
        recs.append(bytes([2]) + x.to_bytes(32, "big"))
        if sq:
            y = pow((x**3 + 7) % P, (P + 1) // 4, P)
            want.append((x, y if y % 2 == 0 else P - y))
        else:
            want.append("bad")
    out, ok = CF.sec1_decompress_batch(np.frombuffer(b"".join(recs), dtype=np.uint8).reshape(-1, 33), nthreads=3)
    for i, w in enumerate(want):
        if w == "bad":
            assert ok[i] == 0 and not out[i].any(), i
        elif w is None:
            assert ok[i] == 1 and not out[i].any(), i
        else:
            assert ok[i] == 1 and out[i].tobytes() == w[0].to_bytes(32, "big") + w[1].to_bytes(32, "big"), i
