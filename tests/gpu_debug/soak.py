"""Soak run (round 5): fresh seeds, every item against the CPU.  For each seed a 2^LOG2-item batch is signed on the GPU (every output array compared with the optimised CPU
leg, oracle/plume_cpu_fast.c, which tests/test_cpu_fast.py holds to the plain oracle), mutated by tests/_fuzz.py and verified in every mode the library has -- V1 with the
first equation in its short form (plume_set_eq1_short 3: any size), in its long form (0), through the checked chain only (2), V2, verify_non_zk V1 / V2, the SEC1 ingest --
from page-locked arrays through the host pipeline AND device-resident; verdicts compared item by item with the CPU's.  The signer runs at uniform levels 1, 0, 2 in turn.
Odd seeds sign from page-locked arrays (round 6: the signer's two-lane uniform pieces), even seeds from pageable ones.
    python3 tests/gpu_debug/soak.py [seeds=4] [log2=18] [first seed=0]"""
import os, sys, pathlib, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import numpy as np, torch
import zk_nullifier_sig_amd as plume
from zk_nullifier_sig_amd import capi
from tests import synth, _fuzz, _cpu_fast as CF

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
log2 = int(sys.argv[2]) if len(sys.argv) > 2 else 18
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = 1 << log2
T = min(64, os.cpu_count() or 1)
eng = plume.Engine(0)
dev = torch.device("cuda:0")
print(eng.version(), "items per batch", n, "cpu threads", T, flush=True)
checked = 0
t_start = time.time()
for sd in range(first, first + seeds):
    start = 900_000_000 + sd * 7_000_003
    b = synth.sign_inputs(n, start=start, seed=0xC0FFEE + sd)
    for ver in (1, 2):
        eng.set_sign_uniform((1, 0, 2)[(sd + ver) % 3])
        src = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")} if sd & 1 else b
        signed = eng.sign_batch(ver, src["msgs"], src["off"], src["sk"], src["r"])
        want_s = CF.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"], nthreads=T)
        for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r"):
            assert np.array_equal(signed[k], want_s[k]), (sd, ver, k, np.nonzero((signed[k] != want_s[k]).any(axis=1))[0][:5])
        checked += n
        v = _fuzz.fuzz_verify_batch(ver, signed, b, seed=1000 * sd + ver)
        rp, hr = (v["r_point"], v["hashed_to_curve_r"]) if ver == 1 else (None, None)
        want = CF.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], rp, hr, nthreads=T)
        assert 0.2 * n < int(want.sum()) < 0.8 * n
        tdev = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
        off = torch.from_numpy(v["off"].view(np.int64)).to(dev)
        for mode in ((3, 0, 2) if ver == 1 else (1,)):
            eng.set_eq1_short(mode)
            got = eng.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], rp, hr)
            assert np.array_equal(got, want), (sd, ver, mode, "host", np.nonzero(got != want)[0][:10])
            ok = torch.zeros(n, dtype=torch.uint8, device=dev)
            eng.verify_batch_device(ver, n, tdev["msgs"], off, int(v["off"][-1]), tdev["pk"], tdev["nullifier"], tdev["c"], tdev["s"], tdev["r_point"] if ver == 1 else None,
                                    tdev["hashed_to_curve_r"] if ver == 1 else None, ok)
            torch.cuda.synchronize()
            assert np.array_equal(ok.cpu().numpy(), want), (sd, ver, mode, "device", np.nonzero(ok.cpu().numpy() != want)[0][:10])
            checked += 2 * n
        eng.set_eq1_short(1)
        z = _fuzz.fuzz_non_zk_batch(ver, signed, b, seed=77 + 1000 * sd + ver)
        want_z = CF.verify_non_zk_batch(ver, z["msgs"], z["off"], z["pk"], z["nullifier"], z["s"], z["r_point"], z["hashed_to_curve_r"], z["c"], nthreads=T)
        for mode in (3, 0):
            eng.set_eq1_short(mode)
            got_z = eng.verify_non_zk_batch(ver, z["msgs"], z["off"], z["pk"], z["nullifier"], z["s"], z["r_point"], z["hashed_to_curve_r"], z["c"])
            assert np.array_equal(got_z, want_z), (sd, ver, mode, "non_zk", np.nonzero(got_z != want_z)[0][:10])
            checked += n
        eng.set_eq1_short(1)
    print(f"seed {sd}: ok   ({checked} item checks so far, {time.time() - t_start:.0f} s)", flush=True)
print("soak ok:", checked, "item checks")
