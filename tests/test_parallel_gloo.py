"""world_size-2 CPU tests (gloo) of the N>1 path: chains are sharded with no data-path collective; the only exchange is
the final all-gather (lsdm_amd/parallel.py).  Each rank's shard is its own reference run with seed base+rank."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from lsdm_amd.parallel import gather_results, gather_samples, shard_range, shard_seed


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_grad_enabled(False)
    torch.set_num_threads(2)
    from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
    from oracle import cpu_ref
    # every rank runs its own chain batch (B_local = 1, K = 1 step) with seed base + rank
    B, K, base = 1, 1, 40
    sd = make_synthetic_state_dict(seed=0)
    bt = make_synthetic_batch(B, seed=shard_seed(base, rank))
    torch.manual_seed(shard_seed(base, rank))
    smp = cpu_ref.Sampler(sd)
    local = smp.sample_loop([B, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"], skip_timesteps=1000 - K)
    # what the caller reads after the loop (run/test_sdm.py:184-202), in one packed collective
    gathered, cat, guid = gather_results(local, smp.saved_cat, smp.saved_guiding_points, dist)
    only = gather_results(local, None, None, dist)
    assert only[1] is None and only[2] is None and torch.equal(only[0], gathered)
    dist.barrier()
    if rank == 0:
        q.put((gathered.numpy(), cat.numpy(), guid.numpy()))
    dist.destroy_process_group()


def test_sharded_chains_allgather_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, cat, guid = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got.shape == (2, 1024, 3) and cat.shape == (2, 1, 13) and guid.shape == (2, 1024, 3)
    # oracle for rank r = the single-process run with seed base + r (no cross-rank coupling)
    torch.set_grad_enabled(False)
    from lsdm_amd.synthetic import make_synthetic_batch, make_synthetic_state_dict
    from oracle import cpu_ref
    sd = make_synthetic_state_dict(seed=0)
    for r in range(world):
        bt = make_synthetic_batch(1, seed=40 + r)
        torch.manual_seed(40 + r)
        smp = cpu_ref.Sampler(sd)
        ref = smp.sample_loop([1, 1024, 3], bt["mask"], bt["given_objs"], bt["given_cats"], bt["text_features"], skip_timesteps=999)
        assert float((torch.from_numpy(got[r]) - ref[0]).abs().max()) <= 1e-5
        assert float((torch.from_numpy(cat[r]) - smp.saved_cat[0]).abs().max()) <= 1e-6
        assert float((torch.from_numpy(guid[r]) - smp.saved_guiding_points[0]).abs().max()) <= 1e-5


def _ragged_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 5
    lo, hi = shard_range(total, rank, world)
    units = torch.arange(total * 6, dtype=torch.float32).view(total, 2, 3)
    got = gather_samples(units[lo:hi], dist, total=total)
    try:        # a shard that disagrees with shard_range is refused on every rank BEFORE the collective (no hang)
        gather_samples(units[:1 + 3 * rank], dist, total=total)
        bad = False
    except ValueError:
        bad = True
    try:        # ... also when only ONE rank is wrong: the sizes are agreed on collectively, so every rank raises
        gather_samples(units[lo:hi] if rank == 0 else units[:1], dist, total=total)
        bad = False
    except ValueError:
        pass
    # fewer units than ranks: rank 1 holds an empty shard, one padded collective all the same
    l1, h1 = shard_range(1, rank, world)
    one = gather_samples(units[l1:h1], dist, total=1)
    assert one.shape == (1, 2, 3) and bool((one == units[:1]).all())
    # the packed collective on a ragged split: three tensors of different widths / dtypes, one all-gather, split back per chain
    cats = torch.arange(total * 4, dtype=torch.float32).view(total, 1, 4) * 0.5
    gpts = -units
    s3, c3, g3 = gather_results(units[lo:hi], cats[lo:hi], gpts[lo:hi], dist, total=total)
    assert torch.equal(s3, units) and torch.equal(c3, cats) and torch.equal(g3, gpts)
    try:
        gather_results(units[lo:hi], cats[:1 + hi - lo], None, dist, total=total)
        raise AssertionError("mismatched chain counts accepted")
    except ValueError:
        pass
    # ... and when only ONE rank holds mismatched / non-fp32 tensors: the failure travels in the size exchange, BOTH ranks raise (no hang)
    for bad_cats in (cats[:1 + hi - lo] if rank == 1 else cats[lo:hi], cats[lo:hi].double() if rank == 0 else cats[lo:hi]):
        try:
            gather_results(units[lo:hi], bad_cats, None, dist, total=total)
            raise AssertionError("a rank with inconsistent tensors went through on rank %d" % rank)
        except ValueError:
            pass
    # ... the same with agree=False (bench.py's timed gather: the sizes were agreed on in the warm-up, ONE collective): the bad rank takes part
    # in that collective with a poisoned block, BOTH ranks raise, nobody waits in a collective the other never issues
    e_lo, e_hi = 2 * rank, 2 * rank + 2           # equal shards of 2 chains
    ok = gather_results(units[e_lo:e_hi], cats[e_lo:e_hi], gpts[e_lo:e_hi], dist, agree=False)
    assert torch.equal(ok[0], units[:4]) and torch.equal(ok[1], cats[:4]) and torch.equal(ok[2], gpts[:4])
    for bad_cats in (cats[:3] if rank == 1 else cats[e_lo:e_hi], cats[e_lo:e_hi].double() if rank == 0 else cats[e_lo:e_hi]):
        try:
            gather_results(units[e_lo:e_hi], bad_cats, gpts[e_lo:e_hi], dist, agree=False)
            raise AssertionError("agree=False: a rank with inconsistent tensors went through on rank %d" % rank)
        except ValueError:
            pass
    dist.barrier()
    if rank == 0:
        q.put((got.numpy(), bad))
    else:
        assert bad
    dist.destroy_process_group()


def test_ragged_shards_allgather_world2():
    """total % world != 0: shard sizes differ by one; one padded all-gather, padding rows dropped, rank order kept."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, bad = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert bad
    np.testing.assert_array_equal(got, np.arange(30, dtype=np.float32).reshape(5, 2, 3))


def test_shard_helpers():
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard_range(512, r, 8) for r in range(8)][-1] == (448, 512)
    assert shard_seed(1, 3) == 4
    x = torch.arange(6.0).view(2, 3)
    assert gather_samples(x, None) is x
