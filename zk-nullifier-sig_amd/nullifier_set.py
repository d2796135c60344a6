"""Nullifier-set post-processing: which items of a verified batch carry a nullifier for the first time?

PLUME exists so that an application can accept ONE nullifier per (public key, message) (reference README.md:5; the field
`PlumeSignature::nullifier`, rust-k256/src/lib.rs:72-73).  The reference stops at `verify`; this module is the step its consumers
run next, on the GPU where the verified nullifiers already are:

    first_occurrence(nullifier, live)                       one GPU, host arrays in / out      (C ABI: plume_nullifier_first_occurrence)
    distributed_first_occurrence(nullifier, live, engine)   every rank holds a shard of the set (the only place this path EXCHANGES data)

Sharded form: a nullifier's owner rank is a function of its bytes, so equal nullifiers meet on one rank.  Records, their global
ids (rank offset + position) and live flags go to their owners with ONE all-to-all (RCCL over xGMI when the tensors are on GPUs:
64 + 8 + 1 bytes per item, n/W items per pair of ranks), the owner marks first occurrences by smallest global id with the same
kernel as the single-GPU form, and one all-to-all of one byte per item brings the flags back.  No other collective except the
sums of the counts."""
import numpy as np

from . import capi


def first_occurrence(nullifier, live=None, engine=None):
    """numpy in / out: (first uint8[n], number of distinct live nullifiers)"""
    eng = engine or capi.default_engine()
    return eng.nullifier_first_occurrence(nullifier, live)


def owner_of(nullifier_t, world: int):
    """rank that owns each 64-byte record: the last two bytes of x, mod world (nullifiers are group elements: uniform)"""
    import torch
    return ((nullifier_t[:, 30].to(torch.int64) << 8) | nullifier_t[:, 31].to(torch.int64)) % world


def distributed_first_occurrence(nullifier_t, live_t, engine, group=None):
    """nullifier_t: torch uint8 [n, 64] (this rank's shard, on the engine's device), live_t: torch uint8 [n] or None.
    Returns (first uint8 [n] on the same device, global number of distinct live nullifiers as int).
    `engine` needs nullifier_first_occurrence_device(n, nullifier, live, ids, first, n_unique) — zk_nullifier_sig_amd.Engine."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = nullifier_t.device
    n = int(nullifier_t.shape[0])
    if live_t is None:
        live_t = torch.ones(n, dtype=torch.uint8, device=dev)
    # global ids: rank offset + position
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[rank] = n
    dist.all_reduce(counts, group=group)
    offset = int(counts[:rank].sum().item())
    ids = torch.arange(n, dtype=torch.int64, device=dev) + offset
    # bucket by owner
    owner = owner_of(nullifier_t, world)
    order = torch.argsort(owner, stable=True)
    send_counts = torch.bincount(owner, minlength=world)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    ssz, rsz = [int(x) for x in send_counts.tolist()], [int(x) for x in recv_counts.tolist()]
    m = sum(rsz)
    rec_in = torch.empty((m, 64), dtype=torch.uint8, device=dev)
    ids_in = torch.empty(m, dtype=torch.int64, device=dev)
    live_in = torch.empty(m, dtype=torch.uint8, device=dev)
    dist.all_to_all_single(rec_in, nullifier_t[order].contiguous(), rsz, ssz, group=group)
    dist.all_to_all_single(ids_in, ids[order].contiguous(), rsz, ssz, group=group)
    dist.all_to_all_single(live_in, live_t[order].contiguous(), rsz, ssz, group=group)
    # owners mark first occurrences among everything they received (smallest global id wins)
    first_in = torch.zeros(m, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    if m:
        engine.nullifier_first_occurrence_device(m, rec_in, live_in, ids_in, first_in, cnt)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
    # flags travel back the way the records came
    flags_back = torch.empty(n, dtype=torch.uint8, device=dev)
    dist.all_to_all_single(flags_back, first_in, ssz, rsz, group=group)
    first = torch.empty(n, dtype=torch.uint8, device=dev)
    first[order] = flags_back
    dist.all_reduce(cnt, group=group)
    return first, int(cnt.item())
