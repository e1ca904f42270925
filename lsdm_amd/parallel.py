"""Multi-GPU sharding of independent diffusion chains.

The reference has no live distributed code.  Chains are independent units, so the job is
partitioned over ranks with NO data-path collective: rank r runs its own reference-equivalent batch of
``B_local`` chains with seed ``base_seed + r``.  The only exchange is one all-gather of the final samples
(``torch.distributed`` backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
import torch


def shard_seed(base_seed, rank):
    """Seed of rank ``rank``'s chain batch: each shard is its own reference run of batch size B_local."""
    return int(base_seed) + int(rank)


def shard_range(total, rank, world):
    """Contiguous [lo, hi) slice of ``total`` units owned by ``rank`` (sizes differ by at most one)."""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_samples(local, dist=None, total=None, agree=True):
    """All-gathers (B_local, ...) tensors into (sum of B_local, ...) in rank order.  ``dist`` is the initialised
    ``torch.distributed`` module or None for a single process.  Equal shard sizes (the bench: 64 chains per rank) take one
    ``all_gather_into_tensor``; with ``total`` = the job's unit count split by ``shard_range`` (sizes differing by one when
    ``total % world != 0``) every rank pads its shard to the largest size for the same single collective and the padding
    rows are dropped afterwards.  ``agree`` (default) first exchanges the shard sizes (one 8-byte all-gather) so that a rank holding
    the wrong shard makes EVERY rank raise instead of leaving the others waiting in the collective; a caller that has already agreed
    on the sizes (bench.py: the untimed warm-up gather) passes ``agree=False`` and issues exactly one collective."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [local.shape[0]] * world if total is None else [shard_range(total, r, world)[1] - shard_range(total, r, world)[0]
                                                            for r in range(world)]
    # validity is agreed on COLLECTIVELY: a rank that raised alone would leave the others waiting in the all-gather
    if agree:
        held = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        every = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(every, held)
        every = [int(v) for v in every.cpu()]
        if every != sizes:
            raise ValueError("shard sizes per rank %s, expected %s (%s)" % (every, sizes, "shard_range(%d)" % total if total is not None
                                                                              else "equal shards: pass total= for a ragged split"))
    big = max(sizes)
    send = local.contiguous()
    if send.shape[0] < big:
        send = torch.cat([send, send.new_zeros((big - send.shape[0],) + tuple(send.shape[1:]))])
    out = torch.empty((world * big,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send)
    if min(sizes) == big:
        return out
    return torch.cat([out[r * big:r * big + n] for r, n in enumerate(sizes)])
