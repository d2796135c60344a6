"""GPU tests added in round 6 (run on the MI355X box: `pytest -m gpu`).

The accept-at-identity input: the signature with nonce r = 0 -- R = Hr = identity, c = SHA256(.. || 00 || 00) mod n, s = c sk -- which the reference's
verify ACCEPTS (rust-k256/src/lib.rs:93-145 has no nonce check; rust-arkworks/src/tests.rs:28-78 neither).  It is the one valid input on which both equations END at the
identity: the short form's accumulator k G - upsilon pk - (tau - 1) R, the long form's last unchecked addition, the half chains' join and V2's 35-byte preimage all meet
their exceptional case on an expected-ACCEPT item.  Here it goes through every form of the verifier the library has; its tamperings ride in the golden edge cases
(tests/golden/make_golden_batches.py) and in the fuzz batches (tests/_fuzz.py plant_nonce_zero), which the older GPU tests consume.
Plus the round's new measurement hooks (plume_get_eq1_short, plume_last_msm_kernel, plume_last_msm_clock) and the small-call rule of plume_set_in_flight.
Everything goes through the C ABI of libplume_hip.so."""
from pathlib import Path

import numpy as np
import pytest

from tests import _fuzz, _sec1
from tests import _oracle_c as OC
from tests import synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def eng():
    import zk_nullifier_sig_amd as plume
    e = plume.Engine(0)
    yield e
    e.close()


def _planted(eng, ver, n, start, every, tamper=False):
    """an honest signed batch with the nonce-zero signature planted at about n / every items; returns (verify arrays, planted indices)"""
    b = synth.sign_inputs(n, start=start)
    sg = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    v = dict(msgs=b["msgs"].copy(), off=b["off"], pk=sg["pk"].copy(), nullifier=sg["nullifier"].copy(), c=sg["c"].copy(), s=sg["s"].copy(), r_point=sg["r_point"].copy(),
             hashed_to_curve_r=sg["hashed_to_curve_r"].copy())
    idx = _fuzz.plant_nonce_zero(ver, v, b, seed=start, every=every, tamper=tamper)
    assert len(idx) >= 1 and not v["r_point"][idx].any() and not v["hashed_to_curve_r"][idx].any()
    return v, idx


def _dev(v, keys):
    import torch
    d = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to("cuda:0") for k in keys}
    d["off"] = torch.from_numpy(v["off"].view(np.int64)).to("cuda:0")
    return d


@pytest.mark.parametrize("ver", [1, 2])
def test_nonce_zero_signature_is_accepted_by_every_form(eng, ver):
    """Untampered plants in an honest batch: every verdict is 1 (= the C oracle's, whole batch), through the long form, the short form forced, the forced fallback, the half
    chains of small calls, verify_non_zk, the SEC1 entry point, host-pointer and device-resident calls, one and eight shards -- and the redo launch sees exactly the tasks
    that end at the identity: both equations of every planted item where one lane walks a chain, none where the halves meet in the (checked) join."""
    import torch
    import zk_nullifier_sig_amd as plume
    n = 20_000                                                                   # above the half chains' threshold (2^14), below the short form's (2^16)
    v, idx = _planted(eng, ver, n, start=31_000_000 + ver, every=16)
    P = len(idx)
    rp, hr = (v["r_point"], v["hashed_to_curve_r"]) if ver == 1 else (None, None)
    want = OC.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], rp, hr, nthreads=16)
    assert bool(want.all()), "the reference accepts the nonce-zero signature"
    d = _dev(v, ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r"))
    mb = int(v["off"][-1])
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    try:
        for mode, kernel, redo in ((0, "k_verify_msm", 2 * P), (3, "k_verify_msm_s", 2 * P), (2, "k_verify_msm_s", n + P)):
            if ver == 2 and mode != 0:
                kernel, redo = "k_verify_msm", 2 * P                             # V2 verify has no R: the long form whatever the mode
            eng.set_eq1_short(mode)
            assert np.array_equal(eng.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], rp, hr), want), ("host pointers", mode)
            ok.zero_()
            eng.verify_batch_device(ver, n, d["msgs"], d["off"], mb, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"] if ver == 1 else None,
                                    d["hashed_to_curve_r"] if ver == 1 else None, ok)
            torch.cuda.synchronize()
            assert bool(ok.all()), ("device-resident", mode)
            assert eng.last_msm_kernel() == kernel and eng.last_redo_tasks() == redo, (mode, eng.last_msm_kernel(), eng.last_redo_tasks(), redo)
            # verify_non_zk (V1 and V2 both check the two equations against the GIVEN identity points and hash .. || 00 || 00)
            nz_kernel, nz_redo = ("k_verify_msm", 2 * P) if mode == 0 else ("k_verify_msm_s", 2 * P if mode == 3 else n + P)
            ok.zero_()
            eng.verify_non_zk_batch_device(ver, n, d["msgs"], d["off"], mb, d["pk"], d["nullifier"], d["s"], d["r_point"], d["hashed_to_curve_r"], d["c"], ok)
            torch.cuda.synchronize()
            assert bool((ok == 1).all()) and eng.last_msm_kernel() == nz_kernel and eng.last_redo_tasks() == nz_redo, (mode, eng.last_redo_tasks(), nz_redo)
        eng.set_eq1_short(1)
        # calls of <= 2^14 items: the chains run as two halves joined by a CHECKED addition -- P == -Q happens in the join, nothing is filed
        m = 5_000
        sub = dict(msgs=v["msgs"], off=v["off"][: m + 1], **{k: v[k][:m] for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")})
        dm = _dev(sub, ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r"))
        okm = torch.zeros(m, dtype=torch.uint8, device="cuda:0")
        eng.verify_batch_device(ver, m, dm["msgs"], dm["off"], int(sub["off"][-1]), dm["pk"], dm["nullifier"], dm["c"], dm["s"], dm["r_point"] if ver == 1 else None,
                                dm["hashed_to_curve_r"] if ver == 1 else None, okm)
        torch.cuda.synchronize()
        assert bool(okm.all()) and eng.last_msm_kernel() == "k_verify_msm_pair" and eng.last_redo_tasks() == 0, (eng.last_msm_kernel(), eng.last_redo_tasks())
        # SEC1 ingest: R and Hr travel as the one-byte-tag records 00
        c33 = {k: _sec1.compress(v[k]) for k in ("pk", "nullifier", "r_point", "hashed_to_curve_r")}
        assert not c33["r_point"][idx].any()
        got = eng.verify_batch_sec1(ver, v["msgs"], v["off"], c33["pk"], c33["nullifier"], v["c"], v["s"], c33["r_point"] if ver == 1 else None, c33["hashed_to_curve_r"] if ver == 1 else None)
        assert np.array_equal(got, want)
    finally:
        eng.set_eq1_short(1)
    # eight shards sharing the GPU (config 4's split): 2 500 items per shard -- every shard runs the half chains
    m8 = plume.Engine([0] * 8)
    try:
        assert np.array_equal(m8.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], rp, hr), want)
        m8.set_eq1_short(3)
        assert np.array_equal(m8.verify_batch(ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], rp, hr), want)
    finally:
        m8.close()


def test_nonce_zero_plants_and_their_tamperings_at_2p17_short_form(eng):
    """a 2^17-item V1 call (the short form by the default rule) salted with nonce-zero plants, half of them tampered in one field, and the fuzz mutations around them: the
    verdicts of the plain C oracle on every planted item and on a sample of the rest; the same batch through verify_non_zk"""
    n = (1 << 17) + 3
    b = synth.sign_inputs(n, start=32_000_000)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = _fuzz.fuzz_verify_batch(1, sg, b, seed=606)
    planted = _fuzz.plant_nonce_zero(1, v, b, seed=607, every=40)               # a second helping at known places (fuzz_verify_batch plants its own with its own seed)
    assert eng.eq1_short() == (1, 1 << 16)
    got = eng.verify_batch(1, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"])
    nz = eng.verify_non_zk_batch(1, v["msgs"], v["off"], v["pk"], v["nullifier"], v["s"], v["r_point"], v["hashed_to_curve_r"], v["c"])
    idx = np.union1d(planted, np.sort(np.random.default_rng(5).choice(n, size=3000, replace=False)))
    msgs = [v["msgs"][int(v["off"][i]):int(v["off"][i + 1])].tobytes() for i in idx]
    sub, off = OC.pack_msgs(msgs)
    a = (sub, off, v["pk"][idx], v["nullifier"][idx])
    want = OC.verify_batch(1, *a, v["c"][idx], v["s"][idx], v["r_point"][idx], v["hashed_to_curve_r"][idx], nthreads=16)
    want_nz = OC.verify_non_zk_batch(1, *a, v["s"][idx], v["r_point"][idx], v["hashed_to_curve_r"][idx], v["c"][idx], nthreads=16)
    assert np.array_equal(got[idx], want) and np.array_equal(nz[idx], want_nz)
    pl = np.searchsorted(idx, planted)
    assert 0 < int(want[pl].sum()) < len(planted), "the plants hold accepted and rejected items"


def test_the_context_says_which_form_and_kernel_ran(eng):
    """plume_get_eq1_short / plume_last_msm_kernel / plume_last_msm_clock: what bench.py names the dominant kernel and prices its cycles with comes from the context, not from
    the environment"""
    import torch
    import zk_nullifier_sig_amd as plume
    assert eng.eq1_short() == (1, 1 << 16)
    n = (1 << 17) + 1
    b = synth.sign_inputs(n, start=33_000_000)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, sg, start=33_000_000)
    d = _dev(v, ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r"))
    exp = synth.expected_ok(n, 33_000_000)
    ok = torch.zeros(n, dtype=torch.uint8, device="cuda:0")

    def call(cnt, ver=1):
        ok.zero_()
        eng.verify_batch_device(ver, cnt, d["msgs"], d["off"], int(v["off"][-1]), d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"] if ver == 1 else None,
                                d["hashed_to_curve_r"] if ver == 1 else None, ok)
        torch.cuda.synchronize()
        return ok[:cnt].cpu().numpy()

    try:
        eng.set_stage_timing(True)
        for cnt, mode, kernel in ((n, 1, "k_verify_msm_s"), (1 << 16, 1, "k_verify_msm_s"), ((1 << 16) - 1, 1, "k_verify_msm"), (1 << 15, 3, "k_verify_msm_s"), (1 << 12, 1, "k_verify_msm_pair"), (1 << 12, 3, "k_verify_msm_s"),
                                  (n, 0, "k_verify_msm")):
            eng.set_eq1_short(mode)
            assert eng.eq1_short()[0] == mode
            assert np.array_equal(call(cnt), exp[:cnt]) and eng.last_msm_kernel() == kernel, (cnt, mode, eng.last_msm_kernel())
            if kernel != "k_verify_msm_pair":
                ghz = eng.last_msm_clock_ghz()
                assert ghz is not None and 0.8 < ghz < 2.6, ghz                 # the MI355X's shader clock tops out at 2.4 GHz
        eng.set_eq1_short(1)
        call(n, ver=2)
        assert eng.last_msm_kernel() == "k_verify_msm"                          # V2 verify: no R, the long form
        eng.set_stage_timing(False)
        call(1 << 16)
        assert eng.last_msm_clock_ghz() is None                                  # sampled on request only
        eng.set_stage_timing(True)
        assert eng.last_msm_clock_ghz() is None                                  # ... and it is the LAST call's sample or nothing: an earlier call's counters are not handed out
    finally:
        eng.set_eq1_short(1)
        eng.set_stage_timing(False)
    fresh = plume.Engine(0)
    try:
        assert fresh.last_msm_kernel() is None
    finally:
        fresh.close()


def test_two_lanes_run_small_calls_side_by_side(eng):
    """plume_set_in_flight(2) with the caller's two streams on different hardware queues (GPU_MAX_HW_QUEUES=8: the Python package sets it before the first HIP call, see
    include/plume_hip.h).  Round 6's trace had shown two torch streams sharing ONE hardware queue under the default pool -- two batches in flight gaining nothing.  Calls of
    2^12 items leave the chip nearly empty: on two streams they must take clearly less than one after the other (measured -31 %; asserted: -12 %), with the verdicts of one
    stream, and every call allocates its lane's workspace (small calls are dealt out to the lanes like large ones)."""
    import os
    import time
    import torch
    if os.environ.get("GPU_MAX_HW_QUEUES") != "8":
        pytest.skip("the process was started with another hardware-queue pool")
    n = 1 << 12
    b = synth.sign_inputs(n, start=34_000_000)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, sg, start=34_000_000)
    d = _dev(v, ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r"))
    exp = torch.from_numpy(synth.expected_ok(n, 34_000_000)).to("cuda:0")
    oks = [torch.zeros(n, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    st = [torch.cuda.Stream(device="cuda:0") for _ in range(2)]
    call = lambda k: eng.verify_batch_device(1, n, d["msgs"], d["off"], int(v["off"][-1]), d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], oks[k], stream=st[k])  # noqa: E731

    def per_call(both, reps=60):
        for _ in range(6):
            call(0); call(1 if both else 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            call(0); call(1 if both else 0)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / (2 * reps)
    try:
        eng.set_in_flight(1)
        one = min(per_call(False) for _ in range(3))
        eng.set_in_flight(2)
        two = min(per_call(True) for _ in range(3))
        assert all(bool((o == exp).all()) for o in oks)
        assert two < 0.88 * one, (one, two)
    finally:
        eng.set_in_flight(1)


def test_signer_host_call_on_two_lanes_uniform_pieces():
    """The signer's host-pointer call from page-locked arrays (round 6): uniform tail-sized pieces dealt to two lanes, twice that size beyond 32 pieces.  Small knobs bring
    both rules within a quick batch (pieces of 2^12 items for 100 001 items, of 2^13 for 300 001, ragged last pieces); the bytes are those of the one-lane tapered call
    and of the pageable call, = the C oracle on a sample."""
    import zk_nullifier_sig_amd as plume
    from zk_nullifier_sig_amd import capi
    OUT = ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")
    e = plume.Engine(0)
    try:
        e.set_host_piece(1 << 15); e.set_host_first_piece(1 << 12); e.set_host_tail_piece(1 << 12)
        for n in (100_001, 300_001):
            b = synth.sign_inputs(n, start=31_000_000 + n)
            pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
            got = {}
            for lanes in (2, 1):
                e.set_host_lanes(lanes)
                so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
                so["status"] = capi.pinned_empty(n)
                so["status"][:] = 0xFF
                r = e.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so)
                assert not r["status"].any()
                got[lanes] = {k: r[k].copy() for k in OUT}
            pageable = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
            for k in OUT:
                assert np.array_equal(got[2][k], got[1][k]) and np.array_equal(got[2][k], pageable[k]), (n, k)
            idx = np.unique(np.concatenate([np.arange(0, n, 997), np.arange(4090, 4100), np.arange(n - 5, n)]))
            msgs = [b["msgs"][b["off"][i]:b["off"][i + 1]].tobytes() for i in idx]
            mb, off = OC.pack_msgs(msgs)
            want = OC.sign_batch(1, mb, off, np.ascontiguousarray(b["sk"][idx]), np.ascontiguousarray(b["r"][idx]), nthreads=8)
            for k in OUT:
                assert np.array_equal(got[2][k][idx], want[k]), (n, k)
    finally:
        e.close()
