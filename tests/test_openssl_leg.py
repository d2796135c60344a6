"""The third-party CPU leg of bench.py's cpu_baseline (oracle/plume_openssl_leg.c: every scalar multiplication, addition and comparison of the verification is OpenSSL
libcrypto's, hash_to_curve and SHA-256 are the oracle's) gives exactly the plain oracle's verdicts: reference-pinned goldens, the edge cases, a fuzzed batch.  An
independent check in both directions: the oracle's group arithmetic against a third party's, on PLUME's own equations."""
import json
import random
from pathlib import Path

import numpy as np
import pytest

from tests import _fuzz, synth
from tests import _openssl_leg as OL
from tests import _oracle_c as OC

pytestmark = pytest.mark.skipif(not OL.available(), reason="libcrypto (headers + library) not available")
GOLD = json.loads((Path(__file__).parent / "golden" / "golden_batches.json").read_text())


def _args(items, ver):
    mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
    return (ver, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32),
            OC.arr(items, "r_point", 64) if ver == 1 else None, OC.arr(items, "hashed_to_curve_r", 64) if ver == 1 else None)


@pytest.mark.parametrize("ver", [1, 2])
def test_goldens_and_edge_cases(ver):
    items = GOLD[f"verify_v{ver}"]
    assert list(OL.verify_batch(*_args(items, ver), nthreads=4)) == [it["ok"] for it in items]
    edge = [e for e in GOLD["edge"] if e["version"] == ver]
    got = OL.verify_batch(*_args(edge, ver), nthreads=2)
    bad = [(it["note"], int(o), it["ok"]) for it, o in zip(edge, got) if int(o) != it["ok"]]
    assert not bad, bad


@pytest.mark.parametrize("ver", [1, 2])
def test_fuzz_vs_plain_oracle(ver):
    n = 1024
    b = synth.sign_inputs(n, start=2_345_678)
    rng = random.Random(ver)
    mb, off = OC.pack_msgs([rng.randbytes(rng.choice([0, 1, 31, 32, 33, 64, 100])) for _ in range(n)])
    signed = OC.sign_batch(ver, mb, off, b["sk"], b["r"], nthreads=8)
    v = _fuzz.fuzz_verify_batch(ver, signed, dict(msgs=mb, off=off), seed=50 + ver)
    args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"] if ver == 1 else None, v["hashed_to_curve_r"] if ver == 1 else None)
    got, want = OL.verify_batch(*args, nthreads=8), OC.verify_batch(*args, nthreads=8)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
    assert 0.2 * n < int(got.sum()) < 0.8 * n
