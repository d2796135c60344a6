"""GPU parity tests added in round 2 (run on the MI355X box: `pytest -m gpu`): plume_arkworks' verify_non_zk on the GPU, BASELINE
config 5 at full size, multi-device contexts, concurrent contexts / streams, page-locked host buffers, error paths of the boundary.
Everything goes through the C ABI of libplume_hip.so; expected values come from the oracles, never from assumptions."""
import json
import threading
from pathlib import Path

import numpy as np
import pytest

from oracle import plume_oracle as O
from tests import _fuzz, synth
from tests import _oracle_c as OC
from tests.test_oracle_c import non_zk_args

pytestmark = pytest.mark.gpu

NONZK = json.loads((Path(__file__).parent / "golden" / "golden_non_zk.json").read_text())["items"]
OUT_KEYS = ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r", "status")


@pytest.fixture(scope="module")
def eng():
    import zk_nullifier_sig_amd as plume
    e = plume.Engine(0)
    yield e
    e.close()


# ------------------------------------------------------------------------------- a9: verify_non_zk (rust-arkworks/src/tests.rs:28-78)
@pytest.mark.parametrize("ver", [1, 2])
def test_verify_non_zk_golden(eng, ver):
    items = [it for it in NONZK if it["version"] == ver]
    ok = eng.verify_non_zk_batch(ver, *non_zk_args(items))
    bad = [(it["note"], int(o), it["ok"]) for it, o in zip(items, ok) if int(o) != it["ok"]]
    assert not bad, bad
    assert {int(o) for o in ok} == {0, 1, 2}


@pytest.mark.parametrize("ver", [1, 2])
def test_verify_non_zk_fuzz_vs_oracle(eng, ver):
    """6144 honest-then-mutated items (zero scalars, identity pk = Err, swapped / negated / foreign points ...): ok[] == C oracle"""
    n = 6144
    b = synth.sign_inputs(n, start=830000)
    signed = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    v = _fuzz.fuzz_non_zk_batch(ver, signed, b, seed=21 + ver)
    args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["s"], v["r_point"], v["hashed_to_curve_r"], v["c"])
    got = eng.verify_non_zk_batch(*args)
    want = OC.verify_non_zk_batch(*args, nthreads=32)
    assert np.array_equal(got, want), [(int(i), int(got[i]), int(want[i])) for i in np.nonzero(got != want)[0][:10]]
    assert {int(x) for x in got} == {0, 1, 2} and 0.15 * n < int((got == 1).sum()) < 0.8 * n


def test_verify_non_zk_checks_both_equations_for_v2(eng):
    """what separates verify_non_zk from PlumeSignature::verify for V2: an honest V2 signature presented with a WRONG r_point / hashed_to_curve_r
    and the digest recomputed over those given points passes the hash comparison but must fail the EC equations (tests.rs:59-70)"""
    n = 64
    b = synth.sign_inputs(n, start=840000)
    sg = eng.sign_batch(2, b["msgs"], b["off"], b["sk"], b["r"])
    assert bool((eng.verify_non_zk_batch(2, b["msgs"], b["off"], sg["pk"], sg["nullifier"], sg["s"], sg["r_point"], sg["hashed_to_curve_r"], sg["c"]) == 1).all())
    rp = np.roll(sg["r_point"], 1, axis=0)                       # somebody else's R
    dg = np.zeros((n, 32), dtype=np.uint8)
    for i in range(n):
        P = O.pt_from_bytes
        d = O.c_hash(2, None, None, P(sg["nullifier"][i].tobytes()), P(rp[i].tobytes()), P(sg["hashed_to_curve_r"][i].tobytes()))
        dg[i] = np.frombuffer((int.from_bytes(d, "big") % O.N).to_bytes(32, "big"), dtype=np.uint8)
    got = eng.verify_non_zk_batch(2, b["msgs"], b["off"], sg["pk"], sg["nullifier"], sg["s"], rp, sg["hashed_to_curve_r"], dg)
    want = OC.verify_non_zk_batch(2, b["msgs"], b["off"], sg["pk"], sg["nullifier"], sg["s"], rp, sg["hashed_to_curve_r"], dg, nthreads=16)
    assert not got.any() and np.array_equal(got, want)


def test_arkworks_facade_sign_then_verify_non_zk(eng, kats):
    """rust-arkworks/src/tests.rs:139-167 (random sign -> verify_non_zk round trip, V1 + V2) and :266-299 (fixed vector) through the façade"""
    import zk_nullifier_sig_amd as plume
    import random
    v = kats["plume_vector"]
    msg = v["msg_utf8"].encode()
    sk = int(v["sk"], 16)
    pk = plume.AffinePoint(int(v["pk_x"], 16), int(v["pk_y"], 16))
    for ver in (plume.PlumeVersion.V1, plume.PlumeVersion.V2):
        sig = plume.sign_with_r((pk, sk), msg, int(v["r"], 16), ver, eng)
        assert plume.verify_non_zk(sig, pk, msg, ver, eng) is True
        assert sig[1].digest_private == int(v["c_v1" if ver == plume.PlumeVersion.V1 else "c_v2"], 16)
        other = plume.PlumeVersion.V2 if ver == plume.PlumeVersion.V1 else plume.PlumeVersion.V1
        assert plume.verify_non_zk(sig, pk, msg, other, eng) is False            # the challenge covers a different preimage
        assert plume.verify_non_zk(sig, pk, msg + b"!", ver, eng) is False
        with pytest.raises(plume.SignatureError):
            plume.verify_non_zk(sig, plume.AffinePoint(), msg, ver, eng)           # Err(HashToCurveError): pk is the identity
    rng = random.Random(5)

    class Rng:
        def fill_bytes(self, k):
            return rng.randbytes(k)

    for _ in range(4):
        sk = rng.randrange(1, O.N)
        pkp = O.pt_mul(sk, O.G)
        pk = plume.AffinePoint(*pkp)
        m = rng.randbytes(rng.randrange(0, 80))
        for ver in (plume.PlumeVersion.V1, plume.PlumeVersion.V2):
            sig = plume.sign(Rng(), (pk, sk), m, ver, eng)
            assert plume.verify_non_zk(sig, pk, m, ver, eng) is True
            pub, prv = sig
            assert O.verify_non_zk(ver.value, m, pkp, (pub.nullifier.x, pub.nullifier.y), pub.s, (prv.r_point.x, prv.r_point.y),
                                   (prv.hashed_to_curve_r.x, prv.hashed_to_curve_r.y), prv.digest_private)


# ------------------------------------------------------------------------------- BASELINE config 5 at full size
def test_config5_arkworks_shape_2p20_identical_to_config3(eng):
    """the same 2^20 batch through the arkworks-shaped entry (pk supplied, not recomputed) gives byte-identical arrays to the
    config-3 run (pk derived), all six outputs + status; a sample is checked against the C oracle, and the whole batch verifies through
    verify_non_zk (the arkworks verification) as well as through verify."""
    n = 1 << 20
    b = synth.sign_inputs(n)
    k256 = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    ark = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"], pk_in=k256["pk"])
    for k in OUT_KEYS:
        assert np.array_equal(k256[k], ark[k]), k
    assert not k256["status"].any()
    idx = np.random.default_rng(3).choice(n, 2048, replace=False)
    sub_msgs = np.concatenate([b["msgs"][32 * i:32 * i + 32] for i in idx] + [np.zeros(16, np.uint8)])
    sub_off = np.arange(len(idx) + 1, dtype=np.uint64) * 32
    want = OC.sign_batch(1, sub_msgs, sub_off, b["sk"][idx], b["r"][idx], pk_in=k256["pk"][idx], nthreads=32)
    for k in OUT_KEYS:
        assert np.array_equal(ark[k][idx], want[k]), k
    ok = eng.verify_non_zk_batch(1, b["msgs"], b["off"], ark["pk"], ark["nullifier"], ark["s"], ark["r_point"], ark["hashed_to_curve_r"], ark["c"])
    assert bool((ok == 1).all())
    ok = eng.verify_batch(1, b["msgs"], b["off"], ark["pk"], ark["nullifier"], ark["c"], ark["s"], ark["r_point"], ark["hashed_to_curve_r"])
    assert bool((ok == 1).all())


# ------------------------------------------------------------------------------- multi-device contexts, threads, streams
def _sign_and_verify(e, ver, b, v=None):
    sg = e.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    vv = synth.corrupt_for_verify(ver, b, sg)
    ok = e.verify_batch(ver, vv["msgs"], vv["off"], vv["pk"], vv["nullifier"], vv["c"], vv["s"], vv.get("r_point"), vv.get("hashed_to_curve_r"))
    return sg, ok


@pytest.mark.parametrize("shards", [2, 3])
def test_multi_device_context_matches_single(eng, shards):
    """plume_init_multi: contiguous even split, one worker thread + streams + staging per shard, results in disjoint slices.
    This box has one GPU, so the shards share device 0 -- same code path as g distinct devices; results must be byte-identical to the
    single-device context, for sizes that do and do not divide evenly, ragged messages included."""
    import zk_nullifier_sig_amd as plume
    m = plume.Engine([0] * shards)
    try:
        assert m.num_shards() == shards and eng.num_shards() == 1
        for n in (1, shards - 1, 1000, 70001):
            b = synth.sign_inputs(n, start=850000)
            for ver in (1, 2):
                want_sg, want_ok = _sign_and_verify(eng, ver, b)
                got_sg, got_ok = _sign_and_verify(m, ver, b)
                for k in OUT_KEYS:
                    assert np.array_equal(got_sg[k], want_sg[k]), (n, ver, k)
                assert np.array_equal(got_ok, want_ok) and list(got_ok) == list(synth.expected_ok(n))
        # ragged messages: shard boundaries fall inside the packed message buffer
        import random
        rng = random.Random(9)
        n = 3001
        msgs = [rng.randbytes(rng.choice([0, 1, 31, 32, 33, 64, 200])) for _ in range(n)]
        mb, off = OC.pack_msgs(msgs)
        b = synth.sign_inputs(n, start=860000)
        a = eng.sign_batch(1, mb, off, b["sk"], b["r"])
        c = m.sign_batch(1, mb, off, b["sk"], b["r"])
        for k in OUT_KEYS:
            assert np.array_equal(a[k], c[k]), k
        assert np.array_equal(m.hash_to_curve_batch(mb, off, a["pk"]), eng.hash_to_curve_batch(mb, off, a["pk"]))
        ok = m.verify_non_zk_batch(1, mb, off, a["pk"], a["nullifier"], a["s"], a["r_point"], a["hashed_to_curve_r"], a["c"])
        assert bool((ok == 1).all())
        # device-resident entry points need a single-device context
        import torch
        t = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
        with pytest.raises(plume.PlumeHipError, match="single-device"):
            m.verify_batch_device(2, 1, t, t, 0, t, t, t, t, None, None, t)
        # an error inside one shard is reported (non-monotonic offsets in the second half only)
        bad = off.copy(); bad[n - 5] = bad[n - 4] + 7
        with pytest.raises(plume.PlumeHipError, match="shard"):
            m.sign_batch(1, mb, bad, b["sk"], b["r"])
        assert np.array_equal(m.sign_batch(1, mb, off, b["sk"], b["r"])["s"], a["s"])      # and the context still works afterwards
    finally:
        m.close()


def test_two_contexts_from_two_threads(eng):
    """distinct contexts are independent: two of them driven concurrently from two host threads on this GPU give the single-context bytes"""
    import zk_nullifier_sig_amd as plume
    n = 40000
    bs = [synth.sign_inputs(n, start=870000 + 100000 * k) for k in range(2)]
    want = [_sign_and_verify(eng, 1, b) for b in bs]
    engs = [plume.Engine(0), plume.Engine(0)]
    got, errs = [None, None], []

    def run(k):
        try:
            for _ in range(3):
                got[k] = _sign_and_verify(engs[k], 1, bs[k])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    [e.close() for e in engs]
    assert not errs, errs
    for k in range(2):
        for key in OUT_KEYS:
            assert np.array_equal(got[k][0][key], want[k][0][key]), (k, key)
        assert np.array_equal(got[k][1], want[k][1])


def test_two_streams_on_one_context(eng):
    """device-resident calls issued back to back on two different streams of ONE context share its workspace; the library orders them
    (hipStreamWaitEvent on the previous call's last kernel), so both give the right bytes"""
    import torch
    dev = torch.device("cuda:0")
    n = 30000
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    batches = []
    for k in range(2):
        b = synth.sign_inputs(n, start=900000 + 50000 * k)
        sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
        v = synth.corrupt_for_verify(1, b, sg)
        batches.append({kk: t(v[kk]) for kk in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")} | {"off": t(v["off"].view(np.int64))})
    oks = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()
    for rep in range(3):
        for k in range(2):
            d = batches[k]
            eng.verify_batch_device(1, n, d["msgs"], d["off"], 32 * n, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], oks[k], stream=streams[k])
    torch.cuda.synchronize()
    exp = synth.expected_ok(n)
    for k in range(2):
        assert np.array_equal(oks[k].cpu().numpy(), exp), k


# ------------------------------------------------------------------------------- page-locked host buffers
def test_pinned_host_buffers_give_the_same_bytes(eng):
    from zk_nullifier_sig_amd import capi
    n = 300000                                              # several pipelined pieces
    b = synth.sign_inputs(n, start=950000)
    want = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
    out = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
    out["status"] = capi.pinned_empty(n)
    got = eng.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=out)
    for k in OUT_KEYS:
        assert np.array_equal(got[k], want[k]), k
    ok = capi.pinned_empty(n)
    eng.verify_batch(1, pin["msgs"], pin["off"], out["pk"], out["nullifier"], out["c"], out["s"], out["r_point"], out["hashed_to_curve_r"], out=ok)
    assert bool(ok.all())
    # per-call registration of pageable arrays (plume_set_host_register_min) changes nothing but the transfer path
    eng.set_host_register_min(1 << 20)
    try:
        got = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    finally:
        eng.set_host_register_min(0)
    for k in OUT_KEYS:
        assert np.array_equal(got[k], want[k]), k
    try:
        for first in (1, 1000, 1 << 17):                    # the first piece's size does not change results
            eng.set_host_first_piece(first)
            assert np.array_equal(eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])["s"], want["s"])
    finally:
        eng.set_host_first_piece(capi.DEFAULT_HOST_FIRST_PIECE)   # the shipped default (plume_capi.hip)


# ------------------------------------------------------------------------------- error paths of the boundary
def test_error_paths(eng):
    import torch
    import zk_nullifier_sig_amd as plume
    dev = torch.device("cuda:0")
    n = 512
    b = synth.sign_inputs(n, start=990000)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    d = {k: t(sg[k]) for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    msgs, off = t(b["msgs"]), t(b["off"].view(np.int64))
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    # n > chunk
    eng.set_chunk(256)
    try:
        with pytest.raises(plume.PlumeHipError, match="chunk"):
            eng.verify_batch_device(1, n, msgs, off, 32 * n, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], ok)
        assert bool(eng.verify_batch(1, b["msgs"], b["off"], sg["pk"], sg["nullifier"], sg["c"], sg["s"], sg["r_point"], sg["hashed_to_curve_r"]).all())  # host path chunks by itself
    finally:
        eng.set_chunk(1 << 20)
    # null arrays / bad version
    with pytest.raises(plume.PlumeHipError, match="null"):
        eng.verify_batch_device(1, n, msgs, off, 32 * n, None, d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], ok)
    with pytest.raises(plume.PlumeHipError, match="V1 needs"):
        eng.verify_batch_device(1, n, msgs, off, 32 * n, d["pk"], d["nullifier"], d["c"], d["s"], None, None, ok)
    with pytest.raises(plume.PlumeHipError, match="version"):
        eng.verify_batch_device(3, n, msgs, off, 32 * n, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], ok)
    # host path: non-monotonic offsets are an argument error
    bad = b["off"].copy(); bad[100] = bad[101] + 1
    with pytest.raises(plume.PlumeHipError, match="non-decreasing"):
        eng.verify_batch(1, b["msgs"], bad, sg["pk"], sg["nullifier"], sg["c"], sg["s"], sg["r_point"], sg["hashed_to_curve_r"])
    # device path: the offsets live on the device, so the KERNELS check them: the item is rejected and msgs is never read out of bounds
    boff = b["off"].copy().view(np.int64)
    boff[100] = 1 << 40                                      # item 99 runs far past the buffer, item 100 "ends" before it starts
    eng.verify_batch_device(1, n, msgs, t(boff), 32 * n, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], ok)
    torch.cuda.synchronize()
    exp = np.ones(n, dtype=np.uint8); exp[99] = exp[100] = 0
    assert np.array_equal(ok.cpu().numpy(), exp)
    eng.verify_batch_device(1, n, msgs, off, 32 * 200 + 5, d["pk"], d["nullifier"], d["c"], d["s"], d["r_point"], d["hashed_to_curve_r"], ok)   # msgs_bytes cuts items 200.. off
    torch.cuda.synchronize()
    assert np.array_equal(ok.cpu().numpy(), (np.arange(n) < 200).astype(np.uint8))
    # sign: flagged through the status byte, outputs still well-formed
    o = {k: torch.zeros((n, w), dtype=torch.uint8, device=dev) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
    st = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng.sign_batch_device(1, n, msgs, t(boff), 32 * n, t(b["sk"]), t(b["r"]), None, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], st)
    torch.cuda.synchronize()
    stn = st.cpu().numpy()
    assert stn[99] == 2 and stn[100] == 2 and int(stn.sum()) == 4
    assert np.array_equal(o["s"].cpu().numpy()[:99], sg["s"][:99])


# ------------------------------------------------------------------------------- f3: circuit witness hints
def test_h2c_intermediates_and_registers(eng, kats):
    """plume_h2c_intermediates_batch + plume_registers_from_be on the GPU / through the C ABI: same checks as the host simulation"""
    import zk_nullifier_sig_amd as plume
    from zk_nullifier_sig_amd import capi
    from tests.test_devsim import _check_h2c_intermediates
    _check_h2c_intermediates(lambda mb, off, pk, regs: eng.h2c_intermediates_batch(mb, off, pk, registers=regs), capi.registers_from_be, kats)
    # device form of the register packing
    import torch
    v = np.frombuffer(np.random.default_rng(1).bytes(32 * 1000), dtype=np.uint8).reshape(1000, 32)
    dv = torch.from_numpy(v.copy()).to("cuda:0")
    out = torch.zeros((1000, 4), dtype=torch.int64, device="cuda:0")
    eng._chk(eng._lib.plume_registers_from_be_device(eng._ctx, 1000, eng._dp(dv), eng._dp(out), None), "plume_registers_from_be_device")
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), capi.registers_from_be(v))
    # the façade's circuit-input dictionary for the reference's fixed vector (circuits/circom/test/v1.test.ts:64-80 uses the same sk / r / message)
    vv = kats["plume_vector"]

    class Mock:
        def fill_bytes(self, n):
            return bytes.fromhex(vv["r"])

    sig = plume.PlumeSignature.sign_v1(plume.SecretKey.from_bytes(bytes.fromhex(vv["sk"])), vv["msg_utf8"].encode(), Mock(), eng)
    ci = plume.circuit_inputs(sig, eng)
    val = lambda regs: sum(int(x) << (64 * t) for t, x in enumerate(regs))  # noqa: E731
    assert val(ci["c"]) == int(vv["c_v1"], 16) and val(ci["s"]) == int(vv["s_v1"], 16)
    assert (val(ci["pk"][0]), val(ci["pk"][1])) == (int(vv["pk_x"], 16), int(vv["pk_y"], 16))
    assert (val(ci["nullifier"][0]), val(ci["nullifier"][1])) == (int(vv["nullifier_x"], 16), int(vv["nullifier_y"], 16))
    u0, u1 = O.hash_to_field2(vv["msg_utf8"].encode() + O.sec1_compress((int(vv["pk_x"], 16), int(vv["pk_y"], 16))))
    assert (val(ci["q0_x_mapped"]), val(ci["q0_y_mapped"])) == O.map_to_curve_sswu(u0)
    assert (val(ci["q1_x_mapped"]), val(ci["q1_y_mapped"])) == O.map_to_curve_sswu(u1)
    # the full input set of plume_v1 (verify_nullifier.circom:14-31): the square-root hints are there too (unpinned definitions, O.sswu_hints restates them)
    assert (val(ci["q0_gx1_sqrt"]), val(ci["q0_gx2_sqrt"]), val(ci["q0_y_pos"])) == O.sswu_hints(u0)
    assert (val(ci["q1_gx1_sqrt"]), val(ci["q1_gx2_sqrt"]), val(ci["q1_y_pos"])) == O.sswu_hints(u1)
    assert set(ci) == {"c", "s", "plume_message", "pk", "nullifier"} | {f"q{k}_{nm}" for k in (0, 1) for nm in ("gx1_sqrt", "gx2_sqrt", "y_pos", "x_mapped", "y_mapped")}


# ------------------------------------------------------------------------------- f2: a non-Python caller of the C ABI
def test_c_program_reproduces_the_reference_vector(tmp_path):
    """tests/abi_c/abi_smoke.c (gcc, only include/plume_hip.h, -lplume_hip): rust-k256/tests/signing.rs:9-21 through the exact FFI surface a Rust
    binding would use -- single device, page-locked buffers, two-shard context, error codes"""
    import subprocess
    from tests.test_abi_cpu import build_abi_smoke
    exe = build_abi_smoke(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "abi_smoke ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_cpp_host_side_passes_the_references_own_tests(tmp_path):
    """tests/abi_cpp/reference_tests.cpp: rust-k256/tests/{signing,verification}.rs and rust-arkworks/src/tests.rs restated test by test against include/plume.hpp
    (C++17, g++, -lplume_hip only) -- the host side a caller of the Rust crates would switch to; plus rejection cases, batch twins and error behaviour"""
    import subprocess
    from tests.test_abi_cpu import build_reference_tests, write_kg_vectors
    exe = build_reference_tests(tmp_path)
    r = subprocess.run([str(exe), str(write_kg_vectors(tmp_path))], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "reference_tests ok" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
    assert "arkworks::test_point_sec1_encoding ... ok" in r.stdout and "signing::test_sign_v1 ... ok" in r.stdout


def test_sec1_der_scalar_marshalling(eng, kats):
    """plume_scalars_to_sec1_der_batch (public keys by the GPU comb) + plume_sec1_der_to_scalars against the wasm README's records"""
    from zk_nullifier_sig_amd import capi
    from tests.test_devsim import _check_sec1_der
    _check_sec1_der(eng.scalars_to_sec1_der_batch, capi.sec1_der_to_scalars, kats)
    # a larger batch against the signer's own generator multiplications: der(sk).public_key == pk
    n = 5000
    b = synth.sign_inputs(n, start=123456)
    sg = eng.sign_batch(2, b["msgs"], b["off"], b["sk"], b["r"])
    der, st = eng.scalars_to_sec1_der_batch(b["sk"])
    assert not st.any() and np.array_equal(der[:, 45:109], sg["pk"]) and np.array_equal(der[:, 7:39], b["sk"])
    # from_sec1_der with the reference's semantics (elliptic-curve validates the embedded public key): records whose key is somebody else's, off the curve,
    # or whose scalar was changed under an unchanged key are rejected; the structure-only host function accepts them (documented difference)
    bad = der.copy()
    bad[1::8, 45:109] = np.roll(der, 1, axis=0)[1::8, 45:109]
    bad[3::8, 108] ^= 1
    bad[5::8, 38] ^= 1
    bad[7::8, 2] ^= 1                                        # framing
    sc, ok = eng.sec1_der_to_scalars(bad)
    want = np.ones(n, dtype=np.uint8); want[1::8] = want[3::8] = want[5::8] = want[7::8] = 0
    assert np.array_equal(ok, want) and np.array_equal(sc[want == 1], b["sk"][want == 1]) and not sc[want == 0].any()
    sc0, ok0 = capi.sec1_der_to_scalars(bad)
    assert ok0[1::8].all() and ok0[3::8].all() and not ok0[7::8].any()


# ------------------------------------------------------------------------------- bench.py contract
def test_bench_line_contract_on_a_small_preset():
    """python bench.py --config 2 (2^16 V1 verify): one JSON line with the contract's keys, the binding roof in `roofline`, the HBM view beside it,
    per-rank values; and the strong-scaling split of config 4's kind (V2) at a size that keeps the test short"""
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    for extra, ver in ((["--config", "2"], 1), (["--scaling", "strong", "--log2-batch", "17", "--version", "2"], 2)):
        r = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-1500:])
        d = json.loads(lines[0])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "hbm_view",
                  "per_rank", "stage_ms"):
            assert k in d, k
        assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "verifies/s" and d["vs_baseline"] is None and f"V{ver}" in d["metric"]
        # the kernel named is the one the context says it launched: equation 1's short form for the 2^16-item V1 call (since round 6: from 2^16 items up), the long form for V2
        assert d["roofline"]["bound"] == "int-valu" and d["roofline"]["kernel"] == ("k_verify_msm_s" if ver == 1 else "k_verify_msm") and 0.05 < d["roofline"]["frac"] < 1.0
        assert d["hbm_view"]["bound"] == "hbm" and d["hbm_view"]["frac"] < 0.05
        assert abs(sum(d["per_rank"]["verifies_per_s"]) - d["value"]) / d["value"] < 1e-6
        assert d["value"] > 1e6


def test_config4_v2_2p22_even_split_over_shards():
    """BASELINE config 4 at its full size: 2^22 V2 verifies split evenly and contiguously over the shards of a multi-device context (this box has one GPU,
    so the shards share it — the sharding code does not care), every item's verdict equal to the corruption pattern, a seeded sample equal to the C oracle,
    and the single-device context agreeing byte for byte"""
    import zk_nullifier_sig_amd as plume
    n = 1 << 22
    b = synth.sign_inputs(n)
    m = plume.Engine([0, 0, 0, 0])
    try:
        sg = m.sign_batch(2, b["msgs"], b["off"], b["sk"], b["r"])
        assert not sg["status"].any()
        v = synth.corrupt_for_verify(2, b, sg)
        ok = m.verify_batch(2, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"])
    finally:
        m.close()
    assert np.array_equal(ok, synth.expected_ok(n))
    idx = np.sort(np.random.default_rng(11).choice(n, size=2048, replace=False))
    sub_msgs = np.concatenate([v["msgs"][32 * i:32 * i + 32] for i in idx] + [np.zeros(16, np.uint8)])
    sub_off = np.arange(len(idx) + 1, dtype=np.uint64) * 32
    want = OC.verify_batch(2, sub_msgs, sub_off, v["pk"][idx], v["nullifier"][idx], v["c"][idx], v["s"][idx], nthreads=16)
    assert np.array_equal(ok[idx], want)
    e = plume.Engine(0)
    try:
        assert np.array_equal(e.verify_batch(2, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"]), ok)
    finally:
        e.close()


def test_bench_under_torchrun_takes_the_rccl_path():
    """the driver's launch line for N > 1 (python -m torch.distributed.run ... bench.py --gpus N) with N = 1 and the RCCL code path forced: process-group init on the
    device, barrier, MAX, gather of the per-rank times -- everything the multi-GPU run uses except a second GPU"""
    import os
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, PLUME_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29641",
                        str(root / "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--log2-batch", "16", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and len(d["per_rank"]["verifies_per_s"]) == 1 and d["value"] > 1e6


@pytest.mark.parametrize("ver", [1, 2])
def test_every_item_of_a_fuzzed_2p20_batch_vs_the_cpu(eng, ver):
    """BASELINE's full size, EVERY item: 2^20 honest-then-mutated signatures (bit flips, garbage records, identities, boundary scalars, negated / foreign / swapped
    points), ok[] compared item by item with the CPU — the optimised CPU leg (oracle/plume_cpu_fast.c), which tests/test_cpu_fast.py holds to the plain oracle's
    verdicts and which is fast enough (≈12 k verifies/s per core) to do the whole batch; a 4096-item sample goes through the plain oracle as well"""
    from tests import _cpu_fast as CF
    import os
    n = 1 << 20
    b = synth.sign_inputs(n, start=5_000_000)
    signed = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
    v = _fuzz.fuzz_verify_batch(ver, signed, b, seed=90 + ver)
    args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"] if ver == 1 else None, v["hashed_to_curve_r"] if ver == 1 else None)
    got = eng.verify_batch(*args)
    threads = min(64, os.cpu_count() or 1)
    want = CF.verify_batch(*args, nthreads=threads)
    assert np.array_equal(got, want), np.nonzero(got != want)[0][:10]
    assert 0.2 * n < int(got.sum()) < 0.8 * n
    idx = np.sort(np.random.default_rng(ver).choice(n, size=4096, replace=False))
    sub_msgs = np.concatenate([v["msgs"][32 * i:32 * i + 32] for i in idx] + [np.zeros(16, np.uint8)])
    sub_off = np.arange(len(idx) + 1, dtype=np.uint64) * 32
    slow = OC.verify_batch(ver, sub_msgs, sub_off, v["pk"][idx], v["nullifier"][idx], v["c"][idx], v["s"][idx],
                           v["r_point"][idx] if ver == 1 else None, v["hashed_to_curve_r"][idx] if ver == 1 else None, nthreads=threads)
    assert np.array_equal(got[idx], slow)
