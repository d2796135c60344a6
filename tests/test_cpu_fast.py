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
