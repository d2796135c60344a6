"""GPU tests added in round 4 (run on the MI355X box: `pytest -m gpu`).

What round 4 changed underneath the C ABI: the field product's column order (tests/test_devsim.py pins the limb arithmetic on the host; the parity tests of rounds
1-3 run unchanged on the GPU), NULL-stream ordering with batches in flight, generator tables built on first use, and `bench.py --gpus 2` run for real as two ranks
on the one GPU of the box.  Everything goes through the C ABI of libplume_hip.so."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

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


def test_null_stream_calls_stay_ordered_with_batches_in_flight(eng):
    """with plume_set_in_flight(ctx, 2) a NULL stream used to resolve to the private stream of whichever lane served the call, so a sign and the verify that
    consumes its outputs -- both issued with stream = NULL -- ran on two unordered streams.  NULL now means the stream of the context the caller holds: the pair is ordered.
    The signer writes into buffers that hold garbage; a verify that overtook it would reject."""
    import torch
    n = 1 << 16
    dev = torch.device("cuda:0")
    b = synth.sign_inputs(n, start=4242)
    msgs, off = torch.from_numpy(b["msgs"]).to(dev), torch.from_numpy(b["off"].view(np.int64)).to(dev)
    sk, r = torch.from_numpy(b["sk"]).to(dev), torch.from_numpy(b["r"]).to(dev)
    nbytes = int(b["off"][-1])
    try:
        eng.set_in_flight(2)
        for rep in range(4):
            o = {f: torch.full((n, w), 0xA5, dtype=torch.uint8, device=dev) for f, w in [("pk", 64), ("nullifier", 64), ("c", 32), ("s", 32), ("r_point", 64), ("hashed_to_curve_r", 64)]}
            status = torch.ones(n, dtype=torch.uint8, device=dev)
            ok = torch.zeros(n, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            eng.sign_batch_device(1, n, msgs, off, nbytes, sk, r, None, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], status)        # lane 0, NULL stream
            eng.verify_batch_device(1, n, msgs, off, nbytes, o["pk"], o["nullifier"], o["c"], o["s"], o["r_point"], o["hashed_to_curve_r"], ok)                       # lane 1, NULL stream
            if rep % 2:   # ... and a third call on the first lane again, reading what the second wrote nothing of: the V2 verdicts of the same signatures differ (c is V1's)
                ok2 = torch.ones(n, dtype=torch.uint8, device=dev)
                eng.verify_batch_device(2, n, msgs, off, nbytes, o["pk"], o["nullifier"], o["c"], o["s"], None, None, ok2)
            torch.cuda.synchronize()
            assert not bool(status.any()), rep
            assert bool(ok.all()), (rep, int((ok == 0).sum()))
            if rep % 2:
                assert not bool(ok2.any())
    finally:
        eng.set_in_flight(1)


@pytest.mark.parametrize("level", [1, 2])
def test_uniform_schedule_signer_every_item_of_2p18(eng, level):
    """plume_set_sign_uniform: the signer with no branch on a digit of sk or r (level 1) and with no table address from one either (level 2) gives the default signer's bytes -- every item of 2^18 with edge scalars salted in (zero-digit
    runs, tiny keys, out-of-range values: status bits), V1 and V2, pk computed and pk supplied -- and a sample of it the C oracle's"""
    import random
    n = 1 << 18
    N = synth.N
    b = synth.sign_inputs(n, start=31_000_000)
    vals = [0, 1, 2, 3, 7, 8, 9, 16, 17, 255, 256, 2**16, 2**64, 2**127, 2**128 - 1, 2**128, 2**128 + 1, 2**192, 2**255, N - 1, N - 2, N // 2, N, N + 1, 2**256 - 1,
            0x1111111111111111111111111111111111111111111111111111111111111111 % N, 0x8888888888888888888888888888888888888888888888888888888888888888 % N]
    rng = random.Random(18)
    idx = rng.sample(range(n), 4096)
    for k, i in enumerate(idx):
        b["sk" if k % 3 else "r"][i] = np.frombuffer(vals[k % len(vals)].to_bytes(32, "big"), dtype=np.uint8)
    for i in rng.sample(range(n), 256):
        b["r"][i] = b["sk"][i]
    try:
        for ver in (1, 2):
            eng.set_sign_uniform(False)
            want = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
            eng.set_sign_uniform(level)
            got = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"])
            for k in want:
                assert np.array_equal(got[k], want[k]), (ver, k)
            assert int((want["status"] != 0).sum()) > 100
            got2 = eng.sign_batch(ver, b["msgs"], b["off"], b["sk"], b["r"], pk_in=want["pk"])
            for k in want:
                assert np.array_equal(got2[k], want[k]), (ver, k, "pk supplied")
        sub = np.sort(np.array(idx[:512] + rng.sample(range(n), 512)))
        sub_msgs = np.concatenate([b["msgs"][32 * i:32 * i + 32] for i in sub] + [np.zeros(16, np.uint8)])
        sub_off = np.arange(len(sub) + 1, dtype=np.uint64) * 32
        ref = OC.sign_batch(2, sub_msgs, sub_off, b["sk"][sub], b["r"][sub], nthreads=8)
        for k in ("pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r", "status"):
            assert np.array_equal(got[k][sub], ref[k]), k
    finally:
        eng.set_sign_uniform(False)


def test_sec1_der_export_follows_the_uniform_levels(eng):
    """the SEC1-DER export multiplies SECRET keys by G: plume_set_sign_uniform covers it (level 1: the comb's uniform schedule, level 2: the scanned table); records identical,
    the level is rejected outside 0..2 and the default is restored"""
    b = synth.sign_inputs(1 << 14, start=77_000)
    sk = b["sk"].copy()
    N = synth.N
    for k, v in enumerate([1, 2, 3, 15, 16, 17, 2**31, 2**32, 2**64 - 1, 2**128, 2**255, N - 1, N - 2, N // 2, (N + 1) // 2]):
        sk[k * 7] = np.frombuffer(v.to_bytes(32, "big"), dtype=np.uint8)
    try:
        eng.set_sign_uniform(0)
        want, st0 = eng.scalars_to_sec1_der_batch(sk)
        assert not st0.any()
        for level in (1, 2):
            eng.set_sign_uniform(level)
            got, st = eng.scalars_to_sec1_der_batch(sk)
            assert np.array_equal(got, want) and np.array_equal(st, st0), level
        with pytest.raises(Exception):
            eng.set_sign_uniform(3)
    finally:
        eng.set_sign_uniform(0)


_LANES = r"""
import json, os, sys
import numpy as np
import zk_nullifier_sig_amd as plume
from zk_nullifier_sig_amd import capi
from tests import synth, _fuzz
n = 300_000
b = synth.sign_inputs(n, start=41_000_000)
eng = plume.Engine(0)
eng.set_stage_timing(True)
eng.set_host_first_piece(1 << 13); eng.set_host_piece(1 << 16)          # 8192, 24576, 65536, 65536, ...: seven pieces, both lanes busy
signed = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
v = _fuzz.fuzz_verify_batch(1, signed, b, seed=404)
pin = {k: capi.pinned_copy(v[k]) for k in ("msgs", "pk", "nullifier", "c", "s", "r_point", "hashed_to_curve_r")}
off = capi.pinned_copy(v["off"])
ok_pinned = eng.verify_batch(1, pin["msgs"], off, pin["pk"], pin["nullifier"], pin["c"], pin["s"], pin["r_point"], pin["hashed_to_curve_r"])
ok_pageable = eng.verify_batch(1, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"])
names = [x for x, _ in eng.last_stage_times()]
np.save(sys.argv[1], np.stack([ok_pinned, ok_pageable]))
print(json.dumps({"lanes": os.environ.get("PLUME_HOST_LANES"), "stages": names, "valid": int(ok_pinned.sum())}))
"""


def test_host_pointer_verify_on_two_lanes_equals_one_lane(tmp_path):
    """Round 4: the pieces of a host-pointer verify alternate between the context and a second lane (four staging slots); pageable caller arrays stay on one lane.  300 000
    fuzzed V1 signatures in seven pieces: the verdicts with PLUME_HOST_LANES=2 (default), =1 and from pageable arrays are the same bytes, and they are the device-resident
    call's (checked against the CPU on a sample)."""
    outs = []
    for lanes in ("2", "1"):
        f = tmp_path / f"ok{lanes}.npy"
        env = dict(os.environ, PYTHONPATH=str(ROOT), PLUME_HOST_LANES=lanes)
        r = subprocess.run([sys.executable, "-c", _LANES, str(f)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["lanes"] == lanes and 0.2 * 300_000 < d["valid"] < 0.9 * 300_000, d
        outs.append(np.load(f))
    assert np.array_equal(outs[0][0], outs[0][1]) and np.array_equal(outs[0], outs[1])
    # a sample against the CPU oracle (the same fuzzed batch, regenerated here)
    import zk_nullifier_sig_amd as plume
    from tests import _fuzz
    n = 300_000
    b = synth.sign_inputs(n, start=41_000_000)
    e = plume.Engine(0)
    try:
        signed = e.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    finally:
        e.close()
    v = _fuzz.fuzz_verify_batch(1, signed, b, seed=404)
    idx = np.sort(np.random.default_rng(4).choice(n, size=1024, replace=False))
    sub_msgs = np.concatenate([v["msgs"][int(v["off"][i]):int(v["off"][i + 1])] for i in idx] + [np.zeros(16, np.uint8)])
    sub_off = np.concatenate([[0], np.cumsum([int(v["off"][i + 1] - v["off"][i]) for i in idx])]).astype(np.uint64)
    want = OC.verify_batch(1, sub_msgs, sub_off, v["pk"][idx], v["nullifier"][idx], v["c"][idx], v["s"][idx], v["r_point"][idx], v["hashed_to_curve_r"][idx], nthreads=8)
    assert np.array_equal(outs[0][0][idx], want)


_LAZY = r"""
import json, sys
import numpy as np
import torch
import zk_nullifier_sig_amd as plume
from tests import synth
MiB = 1 << 20
def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / MiB
torch.zeros(1, device="cuda:0")
u0 = used()
eng = plume.Engine(0)
u1 = used()
b = synth.sign_inputs(256)
sig = eng.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
u2 = used()
ok = eng.verify_batch(1, b["msgs"], b["off"], sig["pk"], sig["nullifier"], sig["c"], sig["s"], sig["r_point"], sig["hashed_to_curve_r"])
u3 = used()
eng.close()
u4 = used()
print(json.dumps({"init": u1 - u0, "sign": u2 - u1, "verify": u3 - u2, "closed": u4 - u0, "ok": int(ok.sum())}))
"""


def test_generator_tables_are_built_by_the_first_call_that_needs_them():
    """plume_init used to allocate and build 1.25 GiB of generator tables whatever the context was for.  Now init takes no table memory, the first sign brings the
    252 MiB comb, the first verify the 1 GiB window table, and closing the last context gives everything back.  (A fresh process: the tables are per process and device.)"""
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    out = subprocess.run([sys.executable, "-c", _LAZY], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["ok"] == 256
    assert d["init"] < 128, d                      # streams, events, a few KiB of scratch
    assert 252 <= d["sign"] < 700, d               # the comb + a small workspace, not the 1 GiB window table
    assert 1024 <= d["verify"] < 1500, d           # the window table
    assert d["closed"] < 256, d                    # the tables and the workspace are back (what stays is the runtime's own: code objects, pools)


def _two_rank_bench(args, timeout=900):
    """bench.py as the driver launches it for N = 2 -- torch.distributed.run, one process per rank -- on a box with ONE GPU: both ranks land on device 0 (LOCAL_RANK modulo
    the visible devices), the timing collectives fall back to gloo (RCCL refuses two ranks on one device).  Launched as a child process before this process's launcher path
    touches anything: the child processes initialise the GPU themselves."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "bench.py", "--gpus", "2"] + args
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                                  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_runs_as_two_ranks_on_the_one_gpu_config4():
    """`bench.py --gpus N>1` had never executed on a GPU box.  Config 4 (V2 verify, the batch split evenly over the ranks) as two ranks sharing device 0:
    world size 2, two per-rank records, the ranks' slices disjoint and covering the batch, every verdict equal to the corruption pattern (checked inside bench.py per rank)."""
    d = _two_rank_bench(["--config", "4", "--log2-batch", "18", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert d["n_gpus"] == 2 and d["world_size"] == 2, d
    assert d["scaling"] == "strong" and d["config"]["version"] == 2, d
    pr = d["ranks"]
    assert len(pr) == 2 and sorted(p["rank"] for p in pr) == [0, 1] and all(p["device"] == 0 for p in pr)
    n = 1 << 18
    spans = sorted((p["slice"][0], p["slice"][1]) for p in pr)
    assert spans == [(0, n // 2), (n // 2, n)], spans                          # [floor(dN/g), floor((d+1)N/g)): disjoint, covering
    assert all(p["verdicts_match_pattern"] for p in pr)
    assert d["timing_backend"] in ("gloo", "nccl")
    assert d["value"] > 0 and d["steps"] == 2


def test_bench_runs_as_two_ranks_on_the_one_gpu_weak_v1():
    """the metric's own workload (V1 verify, weak scaling: 2^16 items per rank here) under the same launcher"""
    d = _two_rank_bench(["--log2-batch", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["scaling"] == "weak"
    assert len(d["ranks"]) == 2 and all(p["verdicts_match_pattern"] for p in d["ranks"]) and len(d["per_rank"]["verifies_per_s"]) == 2
    assert d["config"]["items_per_step_total"] == 2 << 16
