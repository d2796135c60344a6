"""world_size-2 run of the multi-GPU pattern on CPU (gloo): contiguous even split, no data-path collective, barrier +
MAX-over-ranks timing, results landing in disjoint slices.  The per-shard compute is the C oracle (there is no GPU
here); what is under test is bench.py's sharding / aggregation logic."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _worker(rank, world, port, total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import bench
    from tests import _oracle_c as OC
    from tests import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = bench.shard_bounds(total, rank, world)
    b = synth.sign_inputs(hi - lo, start=lo)
    signed = OC.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    v = synth.corrupt_for_verify(1, b, signed, start=lo)
    dist.barrier()
    ok = OC.verify_batch(1, v["msgs"], v["off"], v["pk"], v["nullifier"], v["c"], v["s"], v["r_point"], v["hashed_to_curve_r"])
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    np.save(Path(out_dir) / f"ok_{rank}.npy", ok)
    np.save(Path(out_dir) / f"nul_{rank}.npy", signed["nullifier"])
    dist.destroy_process_group()


def test_two_rank_even_split(tmp_path):
    import bench
    from tests import _oracle_c as OC
    from tests import synth
    total, world = 75, 2     # odd total: shards of 37 and 38
    assert [bench.shard_bounds(total, r, world) for r in range(world)] == [(0, 37), (37, 75)]
    assert [bench.shard_bounds(1 << 22, r, 8)[1] - bench.shard_bounds(1 << 22, r, 8)[0] for r in range(8)] == [1 << 19] * 8
    OC.lib()  # build once before forking
    mp.start_processes(_worker, args=(world, 29611, total, str(tmp_path)), nprocs=world, start_method="fork")
    ok = np.concatenate([np.load(tmp_path / f"ok_{r}.npy") for r in range(world)])
    nul = np.concatenate([np.load(tmp_path / f"nul_{r}.npy") for r in range(world)])
    b = synth.sign_inputs(total)
    ref = OC.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    assert np.array_equal(nul, ref["nullifier"])
    want = synth.expected_ok(total).copy()
    # the shard boundary item 37 is corrupted by kind (37//16)%4 == 2 only if 37 % 16 == 5 -> it is (37 = 2*16+5): its
    # "previous nullifier" lives in the other shard; a shard-local batch leaves it untouched, exactly like item 0 of a batch
    if 37 % 16 == 5 and (37 // 16) % 4 == 2:
        want[37] = 1
    assert np.array_equal(ok, want)


def test_eight_rank_even_split(tmp_path):
    """the same run at world size 8 (the node BASELINE.json's metric is quoted on): eight gloo ranks, slices [floor(dN/8), floor((d+1)N/8)) of a batch that does not divide
    evenly, every rank's verdicts and nullifiers landing in its slice"""
    import bench
    from tests import _oracle_c as OC
    from tests import synth
    total, world = 8 * 11 + 5, 8
    bounds = [bench.shard_bounds(total, r, world) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == total and all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
    assert {hi - lo for lo, hi in bounds} == {11, 12}
    OC.lib()
    mp.start_processes(_worker, args=(world, 29617, total, str(tmp_path)), nprocs=world, start_method="fork")
    ok = np.concatenate([np.load(tmp_path / f"ok_{r}.npy") for r in range(world)])
    nul = np.concatenate([np.load(tmp_path / f"nul_{r}.npy") for r in range(world)])
    b = synth.sign_inputs(total)
    ref = OC.sign_batch(1, b["msgs"], b["off"], b["sk"], b["r"])
    assert np.array_equal(nul, ref["nullifier"])
    want = synth.expected_ok(total).copy()
    for lo, _ in bounds:                       # an item whose corruption is "the previous item's nullifier" and that opens a shard has no previous item there (like item 0 of a batch)
        if lo % 16 == 5 and (lo // 16) % 4 == 2:
            want[lo] = 1
    assert np.array_equal(ok, want)


def test_bench_plans_weak_strong_and_config4():
    """bench.py --scaling weak|strong and --config 4 (BASELINE.json configs[3]: 2^22 V2 verifies, even split): every rank's slice, for every world size the
    driver uses, is contiguous, disjoint, covers the batch, and differs by at most one item between ranks"""
    import bench
    for world in (1, 2, 3, 4, 8):
        for scaling, lg in (("weak", 20), ("strong", 20), ("strong", 22), ("strong", 16)):
            plans = [bench.plan(scaling, lg, world, r) for r in range(world)]
            total = plans[0][0]
            assert total == (1 << lg) * (world if scaling == "weak" else 1) and all(p[0] == total for p in plans)
            assert plans[0][1] == 0 and plans[-1][2] == total
            assert all(plans[r][2] == plans[r + 1][1] for r in range(world - 1))
            sizes = [p[2] - p[1] for p in plans]
            assert max(sizes) - min(sizes) <= 1
            if scaling == "weak":
                assert sizes == [1 << lg] * world
    assert [bench.plan("strong", 22, 8, r)[2] - bench.plan("strong", 22, 8, r)[1] for r in range(8)] == [1 << 19] * 8      # config 4 on 8 GPUs: 2^19 each

    class A:  # the --config presets
        pass
    import sys
    old = sys.argv
    try:
        sys.argv = ["bench.py", "--config", "4", "--gpus", "8"]
        a = bench.parse()
        assert (a.scaling, a.log2_batch, a.version, a.gpus) == ("strong", 22, 2, 8)
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert (a.scaling, a.log2_batch, a.version, a.gpus) == ("weak", 20, 1, 1)
    finally:
        sys.argv = old


def _gather_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # bench.py's aggregation: MAX over ranks for the step time, all_gather for the per-rank report
    mine = torch.tensor([1.0 + 0.25 * rank], dtype=torch.float64)
    got = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(got, mine.clone())
    tmax = mine.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(Path(out_dir) / "agg.npy", np.array([g.item() for g in got] + [tmax.item()]))
    dist.destroy_process_group()


def test_three_rank_timing_aggregation(tmp_path):
    world = 3
    mp.start_processes(_gather_worker, args=(world, 29613, str(tmp_path)), nprocs=world, start_method="fork")
    agg = np.load(tmp_path / "agg.npy")
    assert list(agg) == [1.0, 1.25, 1.5, 1.5]
