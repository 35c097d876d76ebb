"""Sharding a batch of configurations / edges over the GPUs of one node.

The path is embarrassingly parallel: rank r owns the contiguous block [lo, hi) of the batch, the
descriptor (a few KB) is rebuilt identically on every rank, and the ONLY exchange is an all-gather of the
packed validity words (1 bit per configuration: 156 KB per rank for a 1e7 batch on 8 GPUs).  That is
a latency-bound message, so it is one `all_gather_into_tensor` (RCCL over xGMI under the "nccl" backend),
not a bucketed ring schedule.  Shards are multiples of 64 configurations so that every rank's bits start
on a word boundary and the gathered words are the global mask with no repacking.
"""
import numpy as np


def shard_bounds(total: int, world: int, rank: int):
    """[lo, hi) of `rank`: equal shards rounded up to a multiple of 64 (the last ones may be short/empty)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    per = -(-total // world)
    per = -(-per // 64) * 64
    lo = min(total, rank * per)
    hi = min(total, lo + per)
    return lo, hi


def shard_words(total: int, world: int) -> int:
    """Mask words every rank contributes (equal for all ranks, padded with zero bits)."""
    per = -(-total // world)
    return -(-per // 64)


def allgather_mask_words(words, out=None):
    """All-gather each rank's packed int64 mask words; returns the concatenation (world * n_words,)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return words
    if out is None:
        out = torch.empty((world * words.numel(),), dtype=words.dtype, device=words.device)
    if dist.get_backend() == "gloo" and words.is_cuda:
        # rehearsal only (gloo has no CUDA all_gather_into_tensor): stage through the host
        host = torch.empty((world * words.numel(),), dtype=words.dtype)
        dist.all_gather_into_tensor(host, words.cpu().contiguous())
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, words.contiguous())
    return out


def unpack_mask(words, total: int) -> np.ndarray:
    """int64/uint64 words -> (total,) bool (host side, for consumers that want bytes)."""
    w = np.ascontiguousarray(words).view(np.uint64)
    bits = ((w[:, None] >> np.arange(64, dtype=np.uint64)) & np.uint64(1)).astype(bool).reshape(-1)
    return bits[:total]


def sharded_validity(validity_words_fn, q_full, total: int):
    """Run `validity_words_fn(q_shard) -> packed words` on this rank's shard and all-gather the mask.

    `validity_words_fn` is the device call in production (DeviceModel.validity(..., packed=True)); tests
    inject a CPU stand-in to exercise the sharding + collective with the gloo backend.
    Returns the packed words of the WHOLE batch (identical on every rank).
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(total, world, rank)
    n_words = shard_words(total, world)
    words = validity_words_fn(q_full[lo:hi])
    if not torch.is_tensor(words):
        words = torch.from_numpy(np.ascontiguousarray(words).view(np.int64))
    if words.numel() < n_words:          # short / empty last shard: pad with zero bits
        pad = torch.zeros((n_words - words.numel(),), dtype=words.dtype, device=words.device)
        words = torch.cat([words, pad])
    return allgather_mask_words(words)


def pack_bits(flags) -> np.ndarray:
    """(n,) bool -> ceil(n/64) int64 words, bit i%64 of word i//64 (the layout of the device's packed masks)."""
    m = np.ascontiguousarray(flags, dtype=bool)
    pad = (-len(m)) % 64
    bits = np.concatenate([m, np.zeros(pad, dtype=bool)]).reshape(-1, 64)
    return (bits.astype(np.uint64) << np.arange(64, dtype=np.uint64)).sum(axis=1, dtype=np.uint64).view(np.int64)


def sharded_edge_validity(edge_valid_fn, starts, goals, total: int):
    """Edge batches shard exactly like configurations: rank r checks edges [lo, hi) and the E validity bits are
    gathered with the same collective.  `edge_valid_fn(starts, goals) -> (n,) bool` is
    `DiscreteConnector.connect_batch` in production.  Returns the packed words of all E edges on every rank."""
    def words_fn(idx):
        lo, hi = idx
        ok = edge_valid_fn(starts[lo:hi], goals[lo:hi]) if hi > lo else np.zeros((0,), dtype=bool)
        import torch
        if torch.is_tensor(ok):
            ok = ok.cpu().numpy()
        return pack_bits(np.asarray(ok))

    class _Ranges:                                     # lets sharded_validity slice "the batch" into (lo, hi) itself
        def __getitem__(self, s):
            return (s.start, s.stop)
    return sharded_validity(words_fn, _Ranges(), total)


def sharded_records(record_fn, q_full, total: int, width: int):
    """Per-sample float64 records (config 5: distance, pair id, contact points, gradient row ... `width` values per
    sample): rank r computes rows [lo, hi), one all_gather_into_tensor of equal (padded) shards returns all rows."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(total, world, rank)
    per = shard_words(total, world) * 64
    rec = record_fn(q_full[lo:hi]) if hi > lo else np.zeros((0, width))
    if not torch.is_tensor(rec):
        rec = torch.from_numpy(np.ascontiguousarray(rec, dtype=np.float64))
    rec = rec.reshape(-1, width)
    if rec.shape[0] < per:
        rec = torch.cat([rec, torch.zeros((per - rec.shape[0], width), dtype=rec.dtype, device=rec.device)])
    if world == 1:
        return rec[:total]
    out = torch.empty((world * per, width), dtype=rec.dtype, device=rec.device)
    if dist.get_backend() == "gloo" and rec.is_cuda:
        host = torch.empty((world * per, width), dtype=rec.dtype)
        dist.all_gather_into_tensor(host, rec.cpu().contiguous())
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, rec.contiguous())
    # shards are contiguous blocks of `per` rows; drop each shard's padding
    keep = torch.cat([torch.arange(r * per, r * per + (shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0]))
                      for r in range(world)])
    return out[keep.to(out.device)]
