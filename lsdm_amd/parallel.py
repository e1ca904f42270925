"""Multi-GPU sharding of independent diffusion chains.

The reference has no live distributed code.  Chains are independent units, so the job is
partitioned over ranks with NO data-path collective: rank r runs its own reference-equivalent batch of
``B_local`` chains with seed ``base_seed + r``.  The only exchange is one all-gather of the final samples
(``torch.distributed`` backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
import math

import torch

# agree=False (exactly ONE collective): a rank whose local preconditions failed still takes part in that collective, with a block whose first
# column carries this bit pattern (a quiet NaN with a payload no arithmetic produces) -- every rank finds it and raises
_POISON = 0x7FC0DEAD


def shard_seed(base_seed, rank):
    """Seed of rank ``rank``'s chain batch: each shard is its own reference run of batch size B_local."""
    return int(base_seed) + int(rank)


def shard_range(total, rank, world):
    """Contiguous [lo, hi) slice of ``total`` units owned by ``rank`` (sizes differ by at most one)."""
    base, extra = divmod(int(total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_samples(local, dist=None, total=None, agree=True, _invalid=None):
    """All-gathers (B_local, ...) tensors into (sum of B_local, ...) in rank order.  ``dist`` is the initialised
    ``torch.distributed`` module or None for a single process.  Equal shard sizes (the bench: 64 chains per rank) take one
    ``all_gather_into_tensor``; with ``total`` = the job's unit count split by ``shard_range`` (sizes differing by one when
    ``total % world != 0``) every rank pads its shard to the largest size for the same single collective and the padding
    rows are dropped afterwards.  ``agree`` (default) first exchanges the shard sizes (one 8-byte all-gather) so that a rank holding
    the wrong shard makes EVERY rank raise instead of leaving the others waiting in the collective; a caller that has already agreed
    on the sizes (bench.py: the untimed warm-up gather) passes ``agree=False`` and issues exactly one collective."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        if _invalid:
            raise ValueError(_invalid)
        return local
    world = dist.get_world_size()
    sizes = [local.shape[0]] * world if total is None else [shard_range(total, r, world)[1] - shard_range(total, r, world)[0]
                                                            for r in range(world)]
    # validity is agreed on COLLECTIVELY: a rank that raised alone would leave the others waiting in the all-gather
    # (_invalid: a local precondition of the caller that failed -- it travels as a held size of -1, so that every rank raises together)
    if agree or _invalid:
        held = torch.tensor([-1 if _invalid else local.shape[0]], dtype=torch.int64, device=local.device)
        every = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(every, held)
        every = [int(v) for v in every.cpu()]
        if _invalid:
            raise ValueError(_invalid)
        if -1 in every:
            raise ValueError("rank(s) %s hold inconsistent tensors (their own error says which)" % [r for r, v in enumerate(every) if v < 0])
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


def gather_results(sample, saved_cat=None, guiding=None, dist=None, total=None, agree=True):
    """Everything the caller of the path reads after the loop (run/test_sdm.py:184-202: the final sample, ``model.saved_cat`` (B,1,C)
    and, optionally, ``model.saved_guiding_points`` (B,1024,3)) in ONE collective: the per-chain rows of the three
    tensors are packed side by side into one (B_local, 3072 + C [+ 3072]) fp32 buffer, all-gathered once (``gather_samples``: same
    padding and shard-size agreement) and split again.  Returns ``(sample, saved_cat, guiding)`` over all ranks' chains in rank order;
    entries passed as None come back as None.  8 x 64 chains: 6.3 MB with the guiding points -- latency-, not bandwidth-bound.
    A rank whose parts are inconsistent (chain counts, dtype) makes EVERY rank raise ValueError in either mode: with ``agree`` the failure
    travels in the 8-byte size exchange, with ``agree=False`` (the caller vouches for the shard SIZES; one collective, nothing else) inside
    the data all-gather itself as a poisoned block of the agreed shape -- no rank is left waiting in a collective."""
    parts = [("sample", sample)] + [(n, t) for n, t in (("saved_cat", saved_cat), ("guiding", guiding)) if t is not None]
    B = sample.shape[0]
    # local preconditions (every part holds B chains of fp32 values): a failure is AGREED ON with the other ranks inside gather_samples' size
    # exchange -- raising here alone would leave them waiting in the collective
    invalid = None
    for n, t in parts:
        if t.shape[0] != B:
            invalid = "%s holds %d chains, the sample %d" % (n, t.shape[0], B)
        elif t.dtype != torch.float32:
            invalid = "%s is %s: the packed collective carries float32 rows only" % (n, t.dtype)
    single = dist is None or not dist.is_initialized() or dist.get_world_size() == 1
    if single and not invalid:
        return sample, saved_cat, guiding
    if invalid and agree:
        gather_samples(sample.new_zeros((B, 1), dtype=torch.float32), dist, total=total, _invalid=invalid)     # raises on every rank
    if invalid:
        # agree=False: the same single collective as the valid ranks issue, B rows of the packed width, first column = the poison word
        width = sum(math.prod(t.shape[1:]) for _, t in parts)
        mine = torch.zeros((B, width), dtype=torch.float32, device=sample.device)
        mine.view(torch.int32)[:, 0] = _POISON
    else:
        flat = [t.reshape(B, -1) for _, t in parts]
        mine = flat[0] if len(flat) == 1 else torch.cat(flat, dim=1)
    packed = gather_samples(mine, dist, total=total, agree=agree)
    if invalid:
        raise ValueError(invalid)
    if not agree and packed.shape[0] and bool((packed[:, 0].contiguous().view(torch.int32) == _POISON).any()):
        raise ValueError("a rank holds inconsistent tensors (its own error says which)")
    out, col = {}, 0
    for (n, t), f in zip(parts, flat):
        w = f.shape[1]
        out[n] = packed[:, col:col + w].reshape((packed.shape[0],) + tuple(t.shape[1:]))
        col += w
    return out["sample"], out.get("saved_cat"), out.get("guiding")
