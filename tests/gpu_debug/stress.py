"""Randomised stress of the GPU path against the oracles (different seeds each run unless SEED is set): sign / verify batches of ragged
messages with mutated items against the C oracle, and first-occurrence marking against numpy."""
import os, sys, pathlib, time, random
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np
import zk_nullifier_sig_amd as plume
from tests import _oracle_c as OC, synth, _fuzz
seed = int(os.environ.get("SEED", str(int(time.time()))))
print("seed", seed)
rng = random.Random(seed)
engines = [plume.Engine(0), plume.Engine([0, 0]), plume.Engine([0, 0, 0])]     # single device, two and three shards on it
bad = 0
for rnd in range(int(os.environ.get("ROUNDS", "6"))):
    ver = 1 + rnd % 2
    eng = engines[rnd % 3]
    n = rng.choice([1, 63, 64, 65, 1000, 4097, 16384, 70001])
    b = synth.sign_inputs(n, start=rng.randrange(1 << 40))
    msgs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 31, 32, 33, 55, 56, 64, 119, 120, 200]))) for _ in range(n)]
    mb, off = OC.pack_msgs(msgs)
    want = OC.sign_batch(ver, mb, off, b["sk"], b["r"], nthreads=32)
    got = eng.sign_batch(ver, mb, off, b["sk"], b["r"])
    for k in got:
        if not np.array_equal(np.asarray(got[k]).reshape(n, -1), np.asarray(want[k]).reshape(n, -1)):
            print("SIGN MISMATCH", rnd, ver, n, k); bad += 1
    v = _fuzz.fuzz_verify_batch(ver, want, dict(msgs=mb, off=off), seed=rng.randrange(1 << 30))
    args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"] if ver == 1 else None, v["hashed_to_curve_r"] if ver == 1 else None)
    ok = eng.verify_batch(*args)
    wok = OC.verify_batch(*args, nthreads=32)
    if not np.array_equal(ok, wok):
        print("VERIFY MISMATCH", rnd, ver, n, np.nonzero(ok != wok)[0][:8]); bad += 1
    z = _fuzz.fuzz_non_zk_batch(ver, want, dict(msgs=mb, off=off), seed=rng.randrange(1 << 30))
    zargs = (ver, z["msgs"], z["off"], z["pk"], z["nullifier"], z["s"], z["r_point"], z["hashed_to_curve_r"], z["c"])
    zok = eng.verify_non_zk_batch(*zargs)
    zwok = OC.verify_non_zk_batch(*zargs, nthreads=32)
    if not np.array_equal(zok, zwok):
        print("NON_ZK MISMATCH", rnd, ver, n, np.nonzero(zok != zwok)[0][:8]); bad += 1
    print(f"round {rnd}: V{ver} n={n} shards={eng.num_shards()} ok ({int(wok.sum())} valid, non-zk {int((zwok == 1).sum())} true / {int((zwok == 2).sum())} err)")
eng = engines[0]
for rnd in range(int(os.environ.get("DROUNDS", "20"))):
    n = rng.choice([1, 2, 64, 1000, 65536, 1 << 20])
    g = np.random.default_rng(rng.randrange(1 << 30))
    nul = g.integers(0, 256, size=(n, 64), dtype=np.uint8)
    k = max(1, n // rng.choice([2, 3, 16, 1000]))
    src = g.integers(0, n, size=k); dst = g.integers(0, n, size=k)
    nul[dst] = nul[src]
    live = (g.integers(0, 8, size=n) != 0).astype(np.uint8)
    first, cnt = eng.nullifier_first_occurrence(nul, live)
    # numpy definition: first live index of each distinct record
    idx = np.nonzero(live)[0]
    _, inv_first = np.unique(nul[idx].view([("r", "V64")]).reshape(-1), return_index=True)
    want = np.zeros(n, dtype=np.uint8); want[idx[inv_first]] = 1
    if not np.array_equal(first, want) or cnt != int(want.sum()):
        print("DEDUP MISMATCH", rnd, n); bad += 1
print("dedup rounds ok" if not bad else "", "FAILURES:" if bad else "ALL OK", bad)
sys.exit(1 if bad else 0)
