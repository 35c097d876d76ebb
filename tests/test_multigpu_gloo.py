"""The N > 1 path on CPU: world_size-2 gloo.  Sharding + all-gather of the packed validity words; the
per-shard compute is stood in for by the CPU oracle (the product's device call cannot run here)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, total, q, expect_words, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from numbotics_amd.physics import World
        from numbotics_amd.scenes import build_scene
        from numbotics_amd.parallel import sharded_validity, shard_bounds, unpack_mask
        from oracle.cpu_oracle import Oracle
        World()
        arm, chain, obs = build_scene("c2")
        orc = Oracle(arm.scene_model())

        def words_fn(q_shard):
            m = orc.validity(q_shard)
            pad = (-len(m)) % 64
            bits = np.concatenate([m, np.zeros(pad, dtype=bool)]).reshape(-1, 64)
            w = (bits.astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(axis=1, dtype=np.uint64)
            return w.view(np.int64)
        words = sharded_validity(words_fn, q, total)
        lo, hi = shard_bounds(total, world, rank)
        ok = bool(np.array_equal(words.numpy().view(np.uint64), expect_words))
        ok = ok and bool(np.array_equal(unpack_mask(words.numpy(), total), unpack_mask(expect_words, total)))
        ret[rank] = (ok, lo, hi)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [1000, 128, 130])
def test_sharded_mask_allgather_gloo(total, fresh_world):
    from numbotics_amd.scenes import build_scene, sample_q
    from numbotics_amd.parallel import shard_bounds, shard_words
    from oracle.cpu_oracle import Oracle
    arm, chain, obs = build_scene("c2")
    q = sample_q(chain, total, seed=4)
    full = Oracle(arm.scene_model()).validity(q)
    world = 2
    nw = shard_words(total, world)
    per = nw * 64
    padded = np.zeros(world * per, dtype=bool)
    for r in range(world):
        lo, hi = shard_bounds(total, world, r)
        padded[r * per:r * per + (hi - lo)] = full[lo:hi]
    expect = (padded.reshape(-1, 64).astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(axis=1, dtype=np.uint64)
    # with equal 64-aligned shards the gathered words ARE the global mask
    if total % (64 * world) == 0:
        assert np.array_equal(padded[:total], full)
    port = 29500 + (os.getpid() + total) % 2000
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, total, q, expect, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        ok, lo, hi = ret[r]
        assert ok, f"rank {r} gathered a different mask"
    assert ret[0][1] == 0 and ret[world - 1][2] == total


def _worker_edges_records(rank, world, port, E, starts, goals, M, pts, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from numbotics_amd.physics import World
        from numbotics_amd.scenes import build_scene
        from numbotics_amd.parallel import sharded_edge_validity, sharded_records, unpack_mask
        from oracle.cpu_oracle import Oracle
        World()
        arm, chain, obs = build_scene("c2")
        orc = Oracle(arm.scene_model())
        words = sharded_edge_validity(lambda s, g: orc.edge_validity(s, g, 0.05, np.pi, mode="connect")[0], starts, goals, E)

        def records(qs):
            d, idx = orc.closest(qs)
            return np.stack([d, idx.astype(np.float64)], axis=1)
        rec = sharded_records(records, pts, M, 2)
        ret[rank] = (words.numpy().copy(), rec.numpy().copy())
    finally:
        dist.destroy_process_group()


def test_sharded_edges_and_records_gloo(fresh_world):
    """Edge bits and per-sample records (config 3 / config 5 of SURVEY.md 8e) through the same shard + all-gather path."""
    from numbotics_amd.scenes import build_scene, sample_q
    from numbotics_amd.parallel import shard_bounds, shard_words, unpack_mask
    from oracle.cpu_oracle import Oracle
    arm, chain, obs = build_scene("c2")
    orc = Oracle(arm.scene_model())
    E, M, world = 150, 201, 2
    q = sample_q(chain, 2 * E + M, seed=6)
    starts, goals, pts = q[:E], q[E:2 * E], q[2 * E:]
    ok = orc.edge_validity(starts, goals, 0.05, np.pi, mode="connect")[0]
    d, idx = orc.closest(pts)
    port = 31500 + os.getpid() % 2000
    ret = mp.Manager().dict()
    mp.spawn(_worker_edges_records, args=(world, port, E, starts, goals, M, pts, ret), nprocs=world, join=True)
    per = shard_words(E, world) * 64
    for r in range(world):
        words, rec = ret[r]
        bits = unpack_mask(words, world * per)
        got = np.concatenate([bits[k * per:k * per + (shard_bounds(E, world, k)[1] - shard_bounds(E, world, k)[0])] for k in range(world)])
        assert np.array_equal(got, ok)
        assert rec.shape == (M, 2) and np.array_equal(rec[:, 0], d) and np.array_equal(rec[:, 1].astype(np.int32), idx)


def _worker_async(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bufs = [torch.zeros(world * 4, dtype=torch.int64), torch.zeros(world * 4, dtype=torch.int64)]
        pending = [None, None]
        got = []
        for k in range(5):                      # the double-buffered overlap of bench.py, on CPU tensors
            i = k & 1
            if pending[i] is not None:
                pending[i].wait()
                got.append(bufs[i].clone())
            words = torch.full((4,), 100 * k + rank, dtype=torch.int64)
            pending[i] = dist.all_gather_into_tensor(bufs[i], words, async_op=True)
        for i in (1, 0):
            pending[i].wait()
            got.append(bufs[i].clone())
        ret[rank] = [g.tolist() for g in got]
    finally:
        dist.destroy_process_group()


def test_overlapped_gather_pattern_gloo():
    """bench.py overlaps the all-gather of step k with the kernels of step k+1 (two receive buffers, wait before reuse):
    the same call pattern on CPU tensors delivers every step's words."""
    world = 2
    port = 33500 + os.getpid() % 2000
    ret = mp.Manager().dict()
    mp.spawn(_worker_async, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        seen = sorted(tuple(x) for x in ret[r])
        want = sorted(tuple([100 * k + 0] * 4 + [100 * k + 1] * 4) for k in range(5))
        assert seen == want


def test_shard_bounds_cover_and_align():
    from numbotics_amd.parallel import shard_bounds
    for total in (0, 1, 63, 64, 65, 1000, 10_000_000):
        for world in (1, 2, 4, 8):
            prev = 0
            for r in range(world):
                lo, hi = shard_bounds(total, world, r)
                assert lo == prev and lo % 64 == 0 or lo == total
                assert hi >= lo
                prev = hi
            assert prev == total
    lo, hi = shard_bounds(10_000_000, 8, 3)
    assert hi - lo == 1_250_048 - 48 or (hi - lo) % 64 == 0
