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
