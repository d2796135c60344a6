"""GPU parity tests added in round 3 (run on the MI355X box: `pytest -m gpu`).

BASELINE.md §3's criteria taken literally at full size — EVERY item of a 2^20 batch against the CPU, for the signer (config 3: all six output arrays + status), the
SEC1-compressed ingest and plume_arkworks' verify_non_zk — plus the sub-batch overlap of the device-resident calls (results must not depend on it).  The CPU side is
oracle/plume_cpu_fast.c, which tests/test_cpu_fast.py holds to the plain oracle item by item; a sample of every batch goes through the plain oracle as well.
Everything goes through the C ABI of libplume_hip.so."""
import os
import random

import numpy as np
import pytest

from tests import _cpu_fast as CF
from tests import _fuzz, _sec1, synth
from tests import _oracle_c as OC

pytestmark = pytest.mark.gpu
N, P = synth.N, _fuzz.P
OUT = ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r", "status")
THREADS = min(64, os.cpu_count() or 1)


@pytest.fixture(scope="module")
def eng():
    import zk_nullifier_sig_amd as plume
    e = plume.Engine(0)
    yield e
    e.close()


def _rows_differing(a, b):
    a, b = a.reshape(len(a), -1), b.reshape(len(b), -1)
    return np.nonzero((a != b).any(axis=1))[0][:10]


def _salt_sign_inputs(b, n, rng):
    """what the reference's types cannot hold (status bits), tiny / huge keys (pk = +-kG: the checked-addition fallback), r = sk"""
    vals = [0, 1, 2, 3, 7, 8, 9, 16, 17, 255, 256, 257, 32767, 32768, 32769, N - 1, N - 2, N - 8, N - 32768, N, N + 1, 2**256 - 1, 2**255]
    idx = rng.sample(range(n), 4096)
    for k, i in enumerate(idx):
        b["sk" if k % 3 else "r"][i] = np.frombuffer(vals[k % len(vals)].to_bytes(32, "big"), dtype=np.uint8)
    for i in rng.sample(range(n), 512):
        b["r"][i] = b["sk"][i]
    return idx


@pytest.mark.parametrize("ver", [1, 2])
def test_every_item_of_a_2p20_sign_vs_the_cpu(eng, ver):
    """BASELINE config 3 taken literally: 2^20 signs, all six output arrays and the status byte of EVERY item equal to the CPU's (randomizedsigner.rs:43-112)"""
    n = 1 << 20
    b = synth.sign_inputs(n, start=7_000_000)
    salted = _salt_sign_inputs(b, n, random.Random(ver))
    got = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    want = CF.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"], nthreads=THREADS)
    for k in OUT:
        assert np.array_equal(got[k], want[k]), (k, _rows_differing(got[k], want[k]))
    assert int((got["status"] != 0).sum()) > 300
    # the salted items and a random sample through the PLAIN oracle too
    idx = np.sort(np.unique(np.concatenate([np.array(salted[:1024]), np.random.default_rng(ver).choice(n, size=1024, replace=False)])))
    sub_msgs = np.concatenate([b["msgs"][32 * i:32 * i + 32] for i in idx] + [np.zeros(16, np.uint8)])
    sub_off = np.arange(len(idx) + 1, dtype=np.uint64) * 32
    slow = OC.sign_batch(ver, sub_msgs, sub_off, b["sk"][idx], b["r"][idx], nthreads=THREADS)
    for k in OUT:
        assert np.array_equal(got[k][idx], slow[k]), (k, _rows_differing(got[k][idx], slow[k]))


def test_config5_every_item_pk_supplied_2p20(eng):
    """BASELINE config 5 (the arkworks shape: pk supplied, rust-arkworks/src/lib.rs:229-278) at 2^20, every item against the CPU — including supplied keys that are
    somebody else's, the identity, or no curve point"""
    n = 1 << 20
    b = synth.sign_inputs(n, start=9_000_000)
    base = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    pk_in = base["pk"].copy()
    pk_in[7::64] = 0
    pk_in[9::64, 5] ^= 1
    pk_in[11::64] = np.roll(base["pk"], 1, axis=0)[11::64]
    got = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"], pk_in=pk_in)
    want = CF.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"], pk_in=pk_in, nthreads=THREADS)
    for k in OUT:
        assert np.array_equal(got[k], want[k]), (k, _rows_differing(got[k], want[k]))
    honest = np.ones(n, dtype=bool); honest[7::64] = honest[9::64] = honest[11::64] = False
    for k in OUT:
        assert np.array_equal(got[k][honest], base[k][honest]), k


def _fuzz_sec1_records(rec, rng, frac=24):
    """mutations at the level of the 33-byte records: tags, parity, non-canonical x, x with no point, identities"""
    n = len(rec)
    for i in range(n):
        k = rng.randrange(frac)
        if k == 0:
            rec[i, 0] = rng.choice([1, 4, 5, 6, 7, 0x82, 0xFF])
        elif k == 1:
            rec[i, 0] ^= 1                                   # the other root: a valid point, the wrong one
        elif k == 2:
            rec[i, 1:] = np.frombuffer(rng.choice([P, P + 1, 2**256 - 1]).to_bytes(32, "big"), dtype=np.uint8)
        elif k == 3:
            rec[i, 1:] = np.frombuffer(rng.randbytes(32), dtype=np.uint8)   # half of these have no curve point
        elif k == 4:
            rec[i, 0] = 0                                    # identity, junk behind it
        elif k == 5:
            rec[i] = 0


@pytest.mark.parametrize("ver", [1, 2])
def test_every_item_sec1_ingest_2p20_vs_the_cpu(eng, ver):
    """the SEC1-compressed entry point at full size: 2^20 fuzzed signatures presented as 33-byte records, additionally mutated at the record level; expected = the
    reference's semantics, a record that does not deserialize means no signature object, i.e. false (javascript/src/lib.rs:95-118,147-184) — else verify() of the decoded record"""
    n = 1 << 20
    b = synth.sign_inputs(n, start=11_000_000)
    signed = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    v = _fuzz.fuzz_verify_batch(ver, signed, b, seed=70 + ver)
    fields = ["pk", "nullifier"] + (["r_point", "hashed_to_curve_r"] if ver == 1 else [])
    # 64-byte records that are not curve points (garbage, bit flips) have no SEC1 form: such items keep their honest point, the record-level fuzz below covers bad records
    for f in fields:
        dec, okd = CF.sec1_decompress_batch(_sec1.compress(v[f]), nthreads=THREADS)
        bad = (okd == 0) | (dec != v[f]).any(axis=1)
        v[f][bad] = signed[f][bad]
    rec = {f: _sec1.compress(v[f]) for f in fields}
    rng = random.Random(170 + ver)
    for f in fields:
        _fuzz_sec1_records(rec[f], rng, frac=24 * len(fields))
    got = eng.verify_batch_sec1(ver, v["msgs"], v["off"], rec["pk"], rec["nullifier"], v["c"], v["s"], rec.get("r_point"), rec.get("hashed_to_curve_r"))
    dec, decodes = {}, np.ones(n, dtype=bool)
    for f in fields:
        dec[f], okd = CF.sec1_decompress_batch(rec[f], nthreads=THREADS)
        decodes &= okd != 0
    want = CF.verify_batch(ver, v["msgs"], v["off"], dec["pk"], dec["nullifier"], v["c"], v["s"], dec.get("r_point"), dec.get("hashed_to_curve_r"), nthreads=THREADS)
    want = np.where(decodes, want, 0).astype(np.uint8)
    assert np.array_equal(got, want), [(int(i), int(got[i]), int(want[i])) for i in np.nonzero(got != want)[0][:10]]
    assert 0.2 * n < int(got.sum()) < 0.85 * n and int((~decodes).sum()) > 0.02 * n


@pytest.mark.parametrize("ver", [1, 2])
def test_every_item_verify_non_zk_2p20_vs_the_cpu(eng, ver):
    """plume_arkworks' verify_non_zk (rust-arkworks/src/tests.rs:28-78) at full size on a fuzzed batch: honest items, every mutation kind of the verify fuzz, zero scalars,
    identity pk (Err -> 2); every item against the CPU, a sample against the plain oracle"""
    n = 1 << 20
    b = synth.sign_inputs(n, start=13_000_000)
    signed = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    v = _fuzz.fuzz_non_zk_batch(ver, signed, b, seed=80 + ver)
    args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["s"], v["r_point"], v["hashed_to_curve_r"], v["c"])
    got = eng.verify_non_zk_batch(*args)
    want = CF.verify_non_zk_batch(*args, nthreads=THREADS)
    assert np.array_equal(got, want), [(int(i), int(got[i]), int(want[i])) for i in np.nonzero(got != want)[0][:10]]
    assert {int(x) for x in np.unique(got)} == {0, 1, 2} and 0.15 * n < int((got == 1).sum()) < 0.8 * n
    idx = np.sort(np.random.default_rng(ver).choice(n, size=3072, replace=False))
    sub_msgs = np.concatenate([v["msgs"][32 * i:32 * i + 32] for i in idx] + [np.zeros(16, np.uint8)])
    sub_off = np.arange(len(idx) + 1, dtype=np.uint64) * 32
    slow = OC.verify_non_zk_batch(ver, sub_msgs, sub_off, v["pk"][idx], v["nullifier"][idx], v["s"][idx], v["r_point"][idx], v["hashed_to_curve_r"][idx], v["c"][idx], nthreads=THREADS)
    assert np.array_equal(got[idx], slow)


# ------------------------------------------------------------------------------------------- sub-batch overlap (LABNOTES.md §6)
def test_sub_batches_do_not_change_results(eng):
    """device-resident verify (V1, V2, SEC1, non-zk) and sign cut into 1 / 2 / 3 / 4 / 7 / 16 overlapped sub-batches: byte-identical outputs, for sizes that do and do not
    divide, ragged messages across the cuts, rejected items and identities on both sides of a cut"""
    import torch
    from zk_nullifier_sig_amd import capi
    dev = torch.device("cuda:0")
    n = (1 << 17) + 4321
    rng = random.Random(3)
    b = synth.sign_inputs(n, start=15_000_000)
    msgs = [rng.randbytes(rng.choice([0, 1, 31, 32, 33, 64, 100])) for _ in range(n)]
    b["msgs"], b["off"] = OC.pack_msgs(msgs)
    _salt_sign_inputs(b, n, rng)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    d = {k: t(b[k]) for k in ("msgs", "sk", "r")}
    d["off"] = t(b["off"].view(np.int64))
    mb = int(b["off"][-1])
    ref = {}
    try:
        for subs in (1, 2, 3, 4, 7, 16):
            eng.set_sub_batches(subs)
            for ver in (1, 2):
                o = {k: torch.zeros((n, w), dtype=torch.uint8, device=dev) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
                st = torch.zeros(n, dtype=torch.uint8, device=dev)
                eng.sign_batch_device(ver, n, d["msgs"], d["off"], mb, d["sk"], d["r"], None, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], st)
                torch.cuda.synchronize()
                sg = {k: x.cpu().numpy() for k, x in o.items()}
                sg["status"] = st.cpu().numpy()
                if subs == 1:
                    ref["sign", ver] = sg
                    v = _fuzz.fuzz_verify_batch(ver, sg, b, seed=5 + ver)
                    ref["v", ver] = {k: t(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
                    ref["v33", ver] = {k: t(_sec1.compress(sg[k])) for k in ("pk", "nullifier", "r_point", "hashed_to_curve_r")}
                else:
                    for k in OUT:
                        assert np.array_equal(sg[k], ref["sign", ver][k]), (subs, ver, k)
                vd, v33 = ref["v", ver], ref["v33", ver]
                ok = torch.full((n,), 9, dtype=torch.uint8, device=dev)
                eng.verify_batch_device(ver, n, vd["msgs"], d["off"], mb, vd["pk"], vd["nullifier"], vd["c"], vd["s"], vd["r_point"] if ver == 1 else None,
                                        vd["hashed_to_curve_r"] if ver == 1 else None, ok)
                ok2 = torch.full((n,), 9, dtype=torch.uint8, device=dev)
                eng.verify_non_zk_batch_device(ver, n, vd["msgs"], d["off"], mb, vd["pk"], vd["nullifier"], vd["s"], vd["r_point"], vd["hashed_to_curve_r"], vd["c"], ok2)
                ok3 = torch.full((n,), 9, dtype=torch.uint8, device=dev)
                sref = ref["sign", ver]
                eng.verify_batch_sec1_device(ver, n, d["msgs"], d["off"], mb, v33["pk"], v33["nullifier"], t(sref["c"]), t(sref["s"]), v33["r_point"] if ver == 1 else None,
                                             v33["hashed_to_curve_r"] if ver == 1 else None, ok3)
                torch.cuda.synchronize()
                res = (ok.cpu().numpy(), ok2.cpu().numpy(), ok3.cpu().numpy())
                if subs == 1:
                    ref["ok", ver] = res
                    assert 0.2 * n < int(res[0].sum()) < 0.8 * n and set(np.unique(res[1])) == {0, 1, 2}
                    assert bool(res[2][sref["status"] == 0].all())                                 # every signature the reference's types can hold verifies
                else:
                    for a, w in zip(res, ref["ok", ver]):
                        assert np.array_equal(a, w), (subs, ver)
        # and the serial mode's verdicts are the oracle's (a sample)
        idx = np.sort(np.random.default_rng(1).choice(n, size=1024, replace=False))
        v = _fuzz.fuzz_verify_batch(1, ref["sign", 1], b, seed=6)
        mm, oo = OC.pack_msgs([bytes(v["msgs"][int(b["off"][i]):int(b["off"][i + 1])]) for i in idx])
        slow = OC.verify_batch(1, mm, oo, v["pk"][idx], v["nullifier"][idx], v["c"][idx], v["s"][idx], v["r_point"][idx], v["hashed_to_curve_r"][idx], nthreads=THREADS)
        assert np.array_equal(ref["ok", 1][0][idx], slow)
    finally:
        eng.set_sub_batches(capi.DEFAULT_SUB_BATCHES)


def test_stage_times_serial_and_overlapped(eng):
    """plume_last_stage_times: one entry per kernel in the serial mode, one entry for the whole call when sub-batches overlap"""
    import torch
    from zk_nullifier_sig_amd import capi
    dev = torch.device("cuda:0")
    n = 1 << 18
    b = synth.sign_inputs(n, start=16_000_000)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    a = (1, n, t(b["msgs"]), t(b["off"].view(np.int64)), int(b["off"][-1]), t(sg["pk"]), t(sg["nullifier"]), t(sg["c"]), t(sg["s"]), t(sg["r_point"]), t(sg["hashed_to_curve_r"]), ok)
    try:
        eng.set_sub_batches(1)
        eng.set_stage_timing(True)
        eng.verify_batch_device(*a); torch.cuda.synchronize()
        serial = dict(eng.last_stage_times())
        assert list(serial) == ["verify_ingest_h2c", "verify_scalars", "tables", "verify_msm", "verify_finalize"] and bool(ok.all())
        eng.set_sub_batches(4)
        ok.zero_()
        eng.verify_batch_device(*a); torch.cuda.synchronize()
        over = dict(eng.last_stage_times())
        assert list(over) == ["verify_overlapped"] and bool(ok.all())
        assert over["verify_overlapped"] < 1.25 * sum(serial.values())       # (measured: the overlapped order is 1-6 % slower than the serial one, LABNOTES.md §6)
    finally:
        eng.set_sub_batches(capi.DEFAULT_SUB_BATCHES)
        eng.set_stage_timing(False)


# ------------------------------------------------------------------------------------------- multi-GPU readiness
def test_two_distinct_devices_match_one(eng):
    """Engine([0, 1]) against Engine(0): the first use of two DISTINCT device ids.  Skips on the one-GPU boxes of this pool; runs wherever two devices are visible."""
    import torch
    import zk_nullifier_sig_amd as plume
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible devices")
    n = 100_003
    b = synth.sign_inputs(n, start=17_000_000)
    one = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    two = plume.Engine([0, 1])
    try:
        got = two.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
        for k in OUT:
            assert np.array_equal(got[k], one[k]), k
        v = _fuzz.fuzz_verify_batch(1, one, b, seed=9)
        a = (1, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"])
        assert np.array_equal(two.verify_batch(*a), eng.verify_batch(*a))
        assert len(two.shard_numa_nodes()) == 2
    finally:
        two.close()


def test_worker_threads_numa_binding_is_reported_and_harmless(eng):
    """every shard's worker thread is bound to its GPU's NUMA node when sysfs knows it (and left alone otherwise); either way results are the single context's"""
    import zk_nullifier_sig_amd as plume
    n = 20_000
    b = synth.sign_inputs(n, start=18_000_000)
    want = eng.sign_batch(2, b["msgs"], b["off"], b["sk"], b["r"])
    e2 = plume.Engine([0, 0])
    try:
        nodes = e2.shard_numa_nodes()
        assert len(nodes) == 2 and nodes[0] == nodes[1] and nodes[0] >= -1
        got = e2.sign_batch(2, b["msgs"], b["off"], b["sk"], b["r"])
        for k in OUT:
            assert np.array_equal(got[k], want[k]), k
    finally:
        e2.close()
    assert eng.shard_numa_nodes() == [-1]                    # a single-device context has no worker thread


def test_bench_multi_ctx_form_and_config3():
    """bench.py --multi-ctx (one process, plume_init_multi, host-pointer calls from page-locked arrays) and --config 3 (the sign workload) produce contract lines"""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--multi-ctx", "--gpus", "1", "--log2-batch", "17", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "plume_init_multi" in d["config"]["form"] and d["e2e_multi_ctx"]["items_per_s"] == d["value"] and d["value"] > 1e6
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--config", "3", "--log2-batch", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["unit"] == "signs/s" and "configs[2]" in d["config"]["workload"] and d["roofline"]["kernel"] == "k_sign_hmul + k_sign_hdbl" and list(d["stage_ms"])[0] == "sign_gmul"
    parts = d["roofline"]["kernel_ms_parts"]                       # the signer's roofline is priced on both launches of the accounted sk*H, r*H (64 of the 128 doublings moved into k_sign_hdbl)
    assert abs(parts["sign_hmul"] + parts["sign_hdbl"] - d["roofline"]["kernel_ms"]) < 1e-3


# ------------------------------------------------------------------------------------------- circuit hints, the unpinned rest
def test_h2c_hints_unpinned_definitions_on_the_gpu(eng):
    """plume_h2c_hints_batch: q{0,1}_gx1_sqrt / gx2_sqrt / y_pos as include/plume_hip.h defines them (no reference vector exists): the algebraic definitions and the
    Python oracle's independent restatement, through the C ABI; then 2^16 items for internal consistency with the pinned intermediates"""
    from tests.test_devsim import _check_h2c_hints
    _check_h2c_hints(lambda mb, off, pk, regs: eng.h2c_hints_batch(mb, off, pk, regs), lambda mb, off, pk, regs: eng.h2c_intermediates_batch(mb, off, pk, regs))
    n = 1 << 16
    b = synth.sign_inputs(n, start=19_000_000)
    hints, inter = eng.h2c_hints_batch(b["msgs"], b["off"]), eng.h2c_intermediates_batch(b["msgs"], b["off"])
    assert np.array_equal(hints["q0_y_pos"], inter["mapped"][:, 1]) and np.array_equal(hints["q1_y_pos"], inter["mapped"][:, 3])
    assert not (hints["q0_gx1_sqrt"][:, 31] & 1).any() and not (hints["q1_gx2_sqrt"][:, 31] & 1).any()


def test_crafted_items_are_redone_by_the_second_launch_and_cost_only_themselves(eng):
    """items built to hit p == +-q inside an unchecked addition (pk = G, s = +-c = d small: tests/test_devsim.py::test_crafted_collisions_take_the_checked_fallback) are filed
    and redone by k_verify_msm_redo: the verdicts of the whole batch equal the CPU's, honest batches file nothing, every crafted item files its task"""
    import torch
    from oracle import plume_oracle as O
    dev = torch.device("cuda:0")
    n = (1 << 17) + 777
    b = synth.sign_inputs(n, start=21_000_000)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
    msgs, off, ok = t(b["msgs"]), t(b["off"].view(np.int64)), torch.zeros(n, dtype=torch.uint8, device=dev)

    def run(v):                                               # one device-resident call = one multi-scalar launch pair: the counter covers the whole batch
        d = {k: t(v[k]) for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
        eng.verify_batch_device(1, n, msgs, off, int(b["off"][-1]), d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], ok)
        torch.cuda.synchronize()
        return ok.cpu().numpy().copy(), eng.last_redo_tasks()
    got, redone = run(sg)
    assert bool(got.all()) and redone == 0
    v = {k: sg[k].copy() for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    idx = np.arange(5, n, 64)
    d = (idx % 8 + 1).astype(np.uint8)
    v["pk"][idx] = np.frombuffer(O.pt_bytes(O.G), dtype=np.uint8)
    v["c"][idx] = 0; v["c"][idx, 31] = d
    v["s"][idx] = 0; v["s"][idx, 31] = d
    minus = idx[::2]                                          # s = -c: the p == q case (the result is 2d*G), the others p == -q (identity)
    v["s"][minus] = np.frombuffer(b"".join((N - int(x)).to_bytes(32, "big") for x in d[::2]), dtype=np.uint8).reshape(-1, 32)
    want = CF.verify_batch(1, b["msgs"], b["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"], nthreads=THREADS)
    try:
        eng.set_eq1_short(0)                                  # the items are crafted against the LONG form of equation 1 (s G - c pk along one chain of 128 doublings)
        got, redone = run(v)
        # How many of the crafted items meet p == +-q is a property of the digit recoding (rounds 3-4: all of them; with the Eisenstein digits of round 5 the digits of -c no
        # longer mirror the generator's for every d), so the expected count comes from the SAME chain run on the CPU: the device headers compiled for the host
        # (tests/devsim, ds_eq1 -> verify_msm<false>, which counts a fallback exactly when the unchecked chain collided).  The generator's wide digit of a scalar this small
        # is its one bottom digit whatever the window width, so the host build's narrower window does not change the answer.  Equality, not a bound.
        from tests import _devsim as D
        g = O.pt_bytes(O.G)
        hits = {}
        for dd in range(1, 9):
            for neg_s in (False, True):
                before = D.fallback_count()
                D.eq1(((N - dd) if neg_s else dd).to_bytes(32, "big"), dd.to_bytes(32, "big"), g)
                hits[dd, neg_s] = D.fallback_count() - before
                assert hits[dd, neg_s] in (0, 1)
        is_minus = np.zeros(len(idx), dtype=bool); is_minus[::2] = True
        expect = sum(hits[int(x), bool(m)] for x, m in zip(d, is_minus))
        assert 0 < expect <= len(idx) and redone == expect, (redone, expect, len(idx), hits)
        assert np.array_equal(got, want) and int(got.sum()) == n - len(idx)
    finally:
        eng.set_eq1_short(1)
    got, redone = run(v)                                      # round 5: the short form (other coefficients: whatever collides there is filed the same way)
    assert np.array_equal(got, want)
    # ... and crafted against the SHORT form: pk = R, so that the chain adds rows of two equal tables -- equal digits at the top window meet p == q in the first addition
    v2 = {k: sg[k].copy() for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    v2["pk"][idx] = v2["r_point"][idx]
    neg = idx[1::2]                                           # ... or pk = -R: p == -q
    v2["pk"][neg, 32:] = np.frombuffer(b"".join((O.P - int.from_bytes(v2["r_point"][i, 32:].tobytes(), "big")).to_bytes(32, "big") for i in neg), dtype=np.uint8).reshape(-1, 32)
    got, redone = run(v2)
    want = CF.verify_batch(1, b["msgs"], b["off"], v2["pk"], v2["nullifier"], v2["c"], v2["s"], v2["r_point"], v2["hashed_to_curve_r"], nthreads=THREADS)
    assert np.array_equal(got, want) and redone > len(idx) // 64, redone


def test_contexts_of_one_device_share_the_fixed_tables_and_outlive_each_other(eng):
    """The generator's fixed tables are built once per device and process and shared by its contexts (plume_capi.hip FixedTables, reference-counted): a context keeps working
    after the one that built the tables is gone, the last one frees them, and a context created afterwards builds them again -- signatures identical throughout; two contexts
    driven side by side (two batches in flight, what bench.py does) give the verdicts one context gives."""
    import torch
    plume = pytest.importorskip("zk_nullifier_sig_amd")
    n = 3000
    b = synth.sign_inputs(n, start=77)
    want = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    a = plume.Engine(0)
    c = plume.Engine(0)
    got = a.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    a.close()                                              # the module's engine and c still hold the tables
    got2 = c.sign_batch(2, b["msgs"], b["off"], b["sk"], b["r"])
    assert all(np.array_equal(got[k], want[k]) for k in want)
    assert np.array_equal(got2["pk"], want["pk"]) and np.array_equal(got2["nullifier"], want["nullifier"])
    # two contexts, two streams, calls alternating
    v = synth.corrupt_for_verify(1, b, want, start=77)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    off = torch.from_numpy(v["off"].view(np.int64)).to(dev)
    oks = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(2)]
    st = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for i in range(6):
        e = (eng, c)[i % 2]
        e.verify_batch_device(1, n, t["msgs"], off, int(v["off"][-1]), t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], oks[i % 2], stream=st[i % 2])
    torch.cuda.synchronize()
    exp = synth.expected_ok(n, 77)
    assert np.array_equal(oks[0].cpu().numpy(), exp) and np.array_equal(oks[1].cpu().numpy(), exp)
    c.close()


def test_batches_in_flight_do_not_change_results(eng):
    """plume_set_in_flight(ctx, 2): device-resident calls go in turn to two lanes of the context; issued on two streams they run side by side.  Verdicts and signatures are
    those of the one-lane context, the stage times of the last call stay readable, and the knob can be turned back"""
    import torch
    n = 5000
    b = synth.sign_inputs(n, start=901)
    want = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, want, start=901)
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    off = torch.from_numpy(v["off"].view(np.int64)).to(dev)
    sk, r = torch.from_numpy(b["sk"]).to(dev), torch.from_numpy(b["r"]).to(dev)
    msgs_s, off_s = torch.from_numpy(b["msgs"]).to(dev), torch.from_numpy(b["off"].view(np.int64)).to(dev)
    exp = synth.expected_ok(n, 901)
    st = [torch.cuda.Stream(device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    try:
        eng.set_stage_timing(True)                               # (inherited by the lanes set_in_flight creates and by those that exist)
        for k in (2, 3):
            eng.set_in_flight(k)
            oks = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(k)]
            outs = [{f: torch.zeros((n, w), dtype=torch.uint8, device=dev) for f, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]} for _ in range(k)]
            status = [torch.ones(n, dtype=torch.uint8, device=dev) for _ in range(k)]
            for i in range(3 * k):
                j = i % k
                eng.verify_batch_device(1, n, t["msgs"], off, int(v["off"][-1]), t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], oks[j], stream=st[j])
            for i in range(k):
                o = outs[i]
                eng.sign_batch_device(1, n, msgs_s, off_s, int(b["off"][-1]), sk, r, None, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], status[i], stream=st[i])
            torch.cuda.synchronize()
            assert len(eng.last_stage_times()) >= 4
            for j in range(k):
                assert np.array_equal(oks[j].cpu().numpy(), exp), (k, j)
                assert not bool(status[j].any())
                assert all(np.array_equal(outs[j][f].cpu().numpy(), want[f]) for f in outs[j]), (k, j)
        with pytest.raises(Exception):
            eng.set_in_flight(0)
    finally:
        eng.set_in_flight(1)
        eng.set_stage_timing(False)
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng.verify_batch_device(1, n, t["msgs"], off, int(v["off"][-1]), t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], ok)
    torch.cuda.synchronize()
    assert np.array_equal(ok.cpu().numpy(), exp)
