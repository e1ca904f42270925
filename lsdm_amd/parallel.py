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


def gather_samples(local, dist=None):
    """All-gathers (B_local, ...) tensors into (B_local * world, ...) in rank order.  ``dist`` is the initialised
    ``torch.distributed`` module or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out
