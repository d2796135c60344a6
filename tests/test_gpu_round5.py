"""GPU tests added in round 5 (run on the MI355X box: `pytest -m gpu`).

BASELINE.json configs[3] LITERALLY -- 2^22 V2 verifies split evenly over EIGHT shards -- in both forms the library offers: one process with a
plume_init_multi context of eight shards, and `bench.py --gpus 8` as eight torch.distributed ranks.  The pool's boxes have one GPU, so the eight shards / ranks share
device 0; nothing in either form depends on the devices being distinct (eight worker threads, eight workspaces, 8 x 4 staging slots, eight NUMA bindings all exist).
Plus: the signer's default schedule (uniform level 1), the host pipeline's second lane inheriting the context's knobs, mixed page-locked / pageable caller arrays.
Everything goes through the C ABI of libplume_hip.so."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests import _oracle_c as OC
from tests import synth

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
OUT = ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")


@pytest.fixture(scope="module")
def eng():
    import zk_nullifier_sig_amd as plume
    e = plume.Engine(0)
    yield e
    e.close()


def _sample(n, k, seed):
    return np.sort(np.random.default_rng(seed).choice(n, size=k, replace=False))


def _sub_msgs(msgs, idx):
    sub = np.concatenate([msgs[32 * i:32 * i + 32] for i in idx] + [np.zeros(16, np.uint8)])
    return sub, np.arange(len(idx) + 1, dtype=np.uint64) * 32


def test_config4_literal_eight_shards_in_one_context():
    """plume_init_multi([0] * 8): 2^22 V2 signs, then 2^22 V2 verifies of them (1/16 corrupted), from page-locked arrays.  Every verdict = the corruption pattern, a
    2048-item sample = the C oracle, both results = the single-context bytes, eight shards, eight NUMA answers, device memory back after close."""
    import torch
    import zk_nullifier_sig_amd as plume
    from zk_nullifier_sig_amd import capi
    n = 1 << 22
    b = synth.sign_inputs(n)
    pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    m = plume.Engine([0] * 8)
    try:
        assert m.num_shards() == 8
        nodes = m.shard_numa_nodes()
        assert len(nodes) == 8 and all(x >= -1 for x in nodes) and len(set(nodes)) == 1          # eight workers, all bound to device 0's node (or all left alone)
        so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
        so["status"] = capi.pinned_empty(n)
        so["status"][:] = 0xFF
        sg = m.sign_batch(2, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so)
        assert not sg["status"].any()
        v = synth.corrupt_for_verify(2, b, sg)
        vp = {k: capi.pinned_copy(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s")}
        ok = capi.pinned_empty(n)
        ok[:] = 0xFF
        m.verify_batch(2, vp["msgs"], pin["off"], vp["pk"], vp["nullifier"], vp["c"], vp["s"], out=ok)
    finally:
        m.close()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    # eight workspaces (2^19 items each, two lanes), 32 staging slots and the generator tables -- about 40 GB -- are back; what stays is the runtime's own (per-queue scratch for
    # the kernels that spill, code objects), which a second open / use / close cycle must not grow
    assert free0 - free1 < (2 << 30), (free0, free1)
    m = plume.Engine([0] * 8)
    try:
        q = 1 << 20
        ok2 = capi.pinned_empty(q)
        m.verify_batch(2, vp["msgs"], pin["off"][: q + 1], vp["pk"][:q], vp["nullifier"][:q], vp["c"][:q], vp["s"][:q], out=ok2)
        assert np.array_equal(ok2, ok[:q])
    finally:
        m.close()
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info(0)
    assert free1 - free2 < (64 << 20), (free0, free1, free2)
    assert np.array_equal(ok, synth.expected_ok(n))                    # every verdict of the 2^22
    idx = _sample(n, 2048, 55)
    sub, sub_off = _sub_msgs(v["msgs"], idx)
    assert np.array_equal(ok[idx], OC.verify_batch(2, sub, sub_off, v["pk"][idx], v["nullifier"][idx], v["c"][idx], v["s"][idx], nthreads=16))
    subb, subb_off = _sub_msgs(b["msgs"], idx)
    want = OC.sign_batch(2, subb, subb_off, b["sk"][idx], b["r"][idx], nthreads=16)
    for k in OUT:
        assert np.array_equal(sg[k][idx], want[k]), k
    e = plume.Engine(0)
    try:
        assert np.array_equal(e.verify_batch(2, vp["msgs"], pin["off"], vp["pk"], vp["nullifier"], vp["c"], vp["s"]), ok)
        one = e.sign_batch(2, pin["msgs"], pin["off"], pin["sk"], pin["r"])
        for k in OUT + ("status",):
            assert np.array_equal(one[k], sg[k]), k
    finally:
        e.close()


def _ranks_bench(world, args, timeout=1500):
    """bench.py as the driver launches it for N = world (torch.distributed.run, one process per rank) on a box with ONE GPU: every rank lands on device 0 (LOCAL_RANK modulo
    the visible devices) and the timing collectives go over gloo.  Launched as a child process: the ranks initialise the GPU themselves."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           "bench.py", "--gpus", str(world)] + args
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_runs_as_eight_ranks_config4():
    """`bench.py --gpus 8 --config 4 --log2-batch 20` under torch.distributed.run: world size 8, eight disjoint slices [floor(dN/8), floor((d+1)N/8)) covering [0, N), every
    rank's verdicts equal to the corruption pattern (asserted inside bench.py per rank, reported in `ranks`)."""
    d = _ranks_bench(8, ["--config", "4", "--log2-batch", "20", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-probe"])
    assert d["n_gpus"] == 8 and d["world_size"] == 8 and d["scaling"] == "strong" and d["config"]["version"] == 2, d
    pr = d["ranks"]
    assert len(pr) == 8 and sorted(p["rank"] for p in pr) == list(range(8)) and all(p["device"] == 0 for p in pr)
    n = 1 << 20
    spans = sorted((p["slice"][0], p["slice"][1]) for p in pr)
    assert spans == [(n * k // 8, n * (k + 1) // 8) for k in range(8)], spans
    assert all(p["verdicts_match_pattern"] for p in pr)
    assert d["timing_backend"] == "gloo" and d["value"] > 0 and d["steps"] == 2
    assert len(d["per_rank"]["verifies_per_s"]) == 8 and d["config"]["items_per_step_total"] == n


def test_signer_default_is_the_uniform_schedule_level_1():
    """a fresh context signs with plume_set_sign_uniform level 1 (no branch on a digit of sk or r); level 0 is the opt-out.  The stage list names the
    kernels, so the default can be read off a call; bytes are identical either way (every item of 2^16)."""
    import torch
    import zk_nullifier_sig_amd as plume
    n = 1 << 16
    b = synth.sign_inputs(n, start=23_000_000)
    e = plume.Engine(0)
    try:
        assert e.sign_uniform() == 1
        d1 = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
        e.set_sign_uniform(0)
        assert e.sign_uniform() == 0
        d0 = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
        e.set_sign_uniform(1)
    finally:
        e.close()
    for k in OUT + ("status",):
        assert np.array_equal(d0[k], d1[k]), k
    m = 512
    want = OC.sign_batch(1, b["msgs"], b["off"][: m + 1], b["sk"][:m], b["r"][:m], nthreads=8)
    for k in OUT:
        assert np.array_equal(d1[k][:m], want[k]), k
    torch.cuda.synchronize()


def test_host_pipeline_lanes_share_the_contexts_knobs(eng):
    """the host pipeline's second lane did not inherit sign_uniform.  Now every derived context takes every knob from one helper (inherit_tunables).  A 2^19
    host-pointer sign from page-locked arrays at each level, with the context's host pipeline set to two lanes and to one (the signer itself runs on one lane by default:
    round 5: one lane; round 6: two lanes with uniform pieces, PLUME_HOST_SIGN_LANES=1 for the old schedule): identical bytes, = the C oracle on a sample."""
    from zk_nullifier_sig_amd import capi
    n = (1 << 19) + 4097
    b = synth.sign_inputs(n, start=24_000_000)
    pin = {k: capi.pinned_copy(b[k]) for k in ("msgs", "off", "sk", "r")}
    ref = None
    try:
        for level in (1, 0, 2):
            eng.set_sign_uniform(level)
            for lanes in (2, 1):
                eng.set_host_lanes(lanes)
                so = {k: capi.pinned_empty((n, w)) for k, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
                so["status"] = capi.pinned_empty(n)
                so["status"][:] = 0xFF
                got = eng.sign_batch(1, pin["msgs"], pin["off"], pin["sk"], pin["r"], out=so)
                assert not got["status"].any(), (level, lanes)
                if ref is None:
                    ref = {k: got[k].copy() for k in OUT}
                    idx = _sample(n, 1024, 77)
                    sub, sub_off = _sub_msgs(b["msgs"], idx)
                    want = OC.sign_batch(1, sub, sub_off, b["sk"][idx], b["r"][idx], nthreads=16)
                    for k in OUT:
                        assert np.array_equal(ref[k][idx], want[k]), k
                else:
                    for k in OUT:
                        assert np.array_equal(got[k], ref[k]), (level, lanes, k)
    finally:
        eng.set_sign_uniform(1)
        eng.set_host_lanes(2)


def test_one_pageable_array_among_page_locked_ones(eng):
    """the two-lane decision looked at pk and nullifier only.  Calls whose arrays are page-locked except one (msgs, c, the verdict array ...) take the one-lane path
    and give the same verdicts as the all-page-locked and the all-pageable call."""
    from zk_nullifier_sig_amd import capi
    n = 300_001
    b = synth.sign_inputs(n, start=25_000_000)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, sg, start=25_000_000)
    keys = ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")
    vp = {k: capi.pinned_copy(v[k]) for k in keys}
    off = capi.pinned_copy(v["off"])
    want = synth.expected_ok(n, 25_000_000)
    call = lambda a, o: eng.verify_batch(1, a["msgs"], off, a["pk"], a["nullifier"], a["c"], a["s"], a["r_point"], a["hashed_to_curve_r"], out=o)  # noqa: E731
    assert np.array_equal(call(vp, capi.pinned_empty(n)), want)
    for odd in ("msgs", "c", "hashed_to_curve_r"):
        mixed = dict(vp)
        mixed[odd] = v[odd]
        assert np.array_equal(call(mixed, capi.pinned_empty(n)), want), odd
    assert np.array_equal(call(vp, np.empty(n, np.uint8)), want)         # pageable verdict array
    assert np.array_equal(call(v, np.empty(n, np.uint8)), want)


@pytest.mark.parametrize("mode", [3, 2, 0])
def test_equation1_short_form_gives_the_long_forms_verdicts(eng, mode):
    """csrc/plume_eis.h: calls that give R check  k G - upsilon pk - (tau - 1) R == R  instead of  s G - c pk == R.  The golden batch, the 102 edge cases and a fuzzed
    2^17-item batch (V1 verify; verify_non_zk V1 and V2) with the short form forced for every size (3), with every item sent through the scalar stage's fallback (2) and
    with the long form (0): the verdicts of the C oracle each time."""
    import json as _json
    from tests import _fuzz
    gold = _json.loads((ROOT / "tests" / "golden" / "golden_batches.json").read_text())
    try:
        eng.set_eq1_short(mode)
        for items in (gold["verify_v1"], [e for e in gold["edge"] if e["version"] == 1]):
            mb, off = OC.pack_msgs([bytes.fromhex(it["msg"]) for it in items])
            got = eng.verify_batch(1, mb, off, OC.arr(items, "pk", 64), OC.arr(items, "nullifier", 64), OC.arr(items, "c", 32), OC.arr(items, "s", 32), OC.arr(items, "r_point", 64),
                                   OC.arr(items, "hashed_to_curve_r", 64))
            assert [int(x) for x in got] == [it["ok"] for it in items]
        n = (1 << 17) + 5 if mode != 2 else 20_011
        b = synth.sign_inputs(n, start=26_000_000)
        for ver in (1, 2):
            sg = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
            idx = _sample(n, 3000, 91 + ver)
            sub, sub_off = _sub_msgs(b["msgs"], idx)
            if ver == 1:
                v = _fuzz.fuzz_verify_batch(1, sg, b, seed=51)
                got = eng.verify_batch(1, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"])
                vs, vso = _sub_msgs(v["msgs"], idx)
                want = OC.verify_batch(1, vs, vso, v["pk"][idx], v["nullifier"][idx], v["c"][idx], v["s"][idx], v["r_point"][idx], v["hashed_to_curve_r"][idx], nthreads=16)
                assert np.array_equal(got[idx], want) and 0 < int(got.sum()) < n
            nz = eng.verify_non_zk_batch(ver, b["msgs"], b["off"], sg["pk"], sg["nullifier"], sg["s"], sg["r_point"], sg["hashed_to_curve_r"], sg["c"])
            assert bool((nz == 1).all())
            bad = sg["s"].copy()
            bad[idx, 31] ^= 1
            nz = eng.verify_non_zk_batch(ver, b["msgs"], b["off"], sg["pk"], sg["nullifier"], bad, sg["r_point"], sg["hashed_to_curve_r"], sg["c"])
            assert not nz[idx].any() and int(nz.sum()) == n - len(idx)
    finally:
        eng.set_eq1_short(1)


def test_stage_timing_is_opt_in_and_a_refused_call_takes_no_lane_turn(eng):
    """Library 0.5: no timing events between the kernels unless asked for (plume_set_stage_timing) -- plume_last_stage_times then says so instead of returning stale times;
    verdicts are the same either way.  And (found by the CPU pipeline harness, tests/hostsim): with two batches in flight a call REFUSED before it enqueued anything
    (n above the chunk) must not become "the last call" whose stage times are reported."""
    import torch
    import zk_nullifier_sig_amd as plume
    dev = torch.device("cuda:0")
    n = 4096
    b = synth.sign_inputs(n, start=77_000)
    sg = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, sg, start=77_000)
    t = {k: torch.from_numpy(np.ascontiguousarray(v[k])).to(dev) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
    off = torch.from_numpy(v["off"].view(np.int64)).to(dev)
    exp = synth.expected_ok(n, 77_000)
    call = lambda ok, cnt=n: eng.verify_batch_device(1, cnt, t["msgs"], off, int(v["off"][-1]), t["pk"], t["nullifier"], t["c"], t["s"], t["r_point"], t["hashed_to_curve_r"], ok)  # noqa: E731
    ok0, ok1 = (torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(2))
    call(ok0); torch.cuda.synchronize()
    with pytest.raises(plume.PlumeHipError, match="stage timing is off"):
        eng.last_stage_times()
    try:
        eng.set_stage_timing(True)
        eng.set_in_flight(2)
        call(ok1); torch.cuda.synchronize()
        first = eng.last_stage_times()
        assert [s for s, _ in first] == ["verify_ingest_h2c+scalars", "tables", "verify_msm", "verify_finalize"] and all(ms > 0 for _, ms in first)
        eng.set_chunk(1024)
        with pytest.raises(plume.PlumeHipError, match="chunk"):
            call(ok1)                                                   # refused: 4096 items, chunk 1024
        assert eng.last_stage_times() == first                        # still the call that ran, on the lane that ran it
    finally:
        eng.set_chunk(1 << 20)
        eng.set_in_flight(1)
        eng.set_stage_timing(False)
    assert np.array_equal(ok0.cpu().numpy(), exp) and np.array_equal(ok1.cpu().numpy(), exp)


def test_ragged_calls_through_randomly_configured_contexts():
    """tests/gpu_debug/soak_ragged.py, sixty iterations: calls of 1 .. 65 537 items through contexts with randomly chosen piece sizes, chunk, lanes, first-equation form,
    signer level and one to five shards sharing the GPU; every signer output and every verdict of both verify semantics against the CPU (oracle/plume_cpu_fast.c).
    (A child process: the script owns its engines.  Round 5 ran it for 2 900 iterations: profiles/r05_soak.txt.)"""
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "gpu_debug" / "soak_ragged.py"), "60", "5"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "ragged soak ok: 60 iterations" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_half_chains_of_small_calls_give_the_one_lane_verdicts():
    """calls of at most 2^14 items run every long-form chain of the multi-scalar stage as two half chains on two lanes (k_verify_msm_pair): the verdicts are those of a
    context that never does (PLUME_MSM_PAIR_MAX=0) and the CPU's -- on a fuzzed batch salted with the crafted items whose slots meet in p == +-q (pk = G, s = +-c small:
    with half chains the generator's slot and pk's meet in the JOIN, a checked addition, so nothing of theirs is filed any more) and with items whose two slots are the same
    point (nullifier = H-multiple games: pk = nullifier), V1 and V2, several sizes around the threshold"""
    import zk_nullifier_sig_amd as plume
    from oracle import plume_oracle as O
    from tests import _cpu_fast as CF
    from tests import _fuzz
    os.environ["PLUME_MSM_PAIR_MAX"] = "0"
    try:
        one = plume.Engine(0)
    finally:
        del os.environ["PLUME_MSM_PAIR_MAX"]
    two = plume.Engine(0)
    try:
        for ver, n in ((1, 5000), (2, 3001), (1, 16384), (1, 16385), (2, 1), (1, 130)):
            b = synth.sign_inputs(n, start=88_000_000 + n)
            sg = two.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
            v = _fuzz.fuzz_verify_batch(ver, sg, b, seed=n)
            idx = np.arange(3, n, 17)
            d = (idx % 8 + 1).astype(np.uint8)
            v["pk"][idx] = np.frombuffer(O.pt_bytes(O.G), dtype=np.uint8)
            v["c"][idx] = 0; v["c"][idx, 31] = d
            v["s"][idx] = 0; v["s"][idx, 31] = d
            minus = idx[::2]
            if len(minus):
                v["s"][minus] = np.frombuffer(b"".join((O.N - int(x)).to_bytes(32, "big") for x in d[::2]), dtype=np.uint8).reshape(-1, 32)
            same = np.arange(7, n, 29)
            v["nullifier"][same] = v["pk"][same]
            args = (ver, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"] if ver == 1 else None, v["hashed_to_curve_r"] if ver == 1 else None)
            want = OC.verify_batch(*args, nthreads=min(32, os.cpu_count() or 1))           # the PLAIN oracle (0.35 ms per item and thread: these batches are small)
            assert np.array_equal(CF.verify_batch(*args, nthreads=min(32, os.cpu_count() or 1)), want)
            for eq1 in (0, 1):
                one.set_eq1_short(eq1); two.set_eq1_short(eq1)
                assert np.array_equal(one.verify_batch(*args), want), (ver, n, eq1, "one lane per chain")
                assert np.array_equal(two.verify_batch(*args), want), (ver, n, eq1, "half chains")
    finally:
        one.close(); two.close()
