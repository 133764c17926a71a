"""Multi-GPU plumbing of the generation path: one process per GPU, utterances sharded statically, and exactly ONE
collective — a bucketed broadcast of the weights from rank 0 at start-up (RCCL over xGMI on MI355X).
Steady state has no collective: the reference's inference path has none either, utterances are
independent, and each rank returns its own waveforms. Backend-agnostic (``nccl`` = RCCL on ROCm, ``gloo`` on CPU
for the world_size-2 tests).
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced static partition: the first ``n_items % world`` ranks get one extra item."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors: Sequence[torch.Tensor], rank: int, world: int) -> List[torch.Tensor]:
    """Slice every [batch, ...] tensor to this rank's utterances (None entries pass through)."""
    n = next(t.shape[0] for t in tensors if t is not None)
    lo, hi = shard_range(n, rank, world)
    return [t[lo:hi] if t is not None else None for t in tensors]


def broadcast_tensors(tensors: Iterable[torch.Tensor], src: int = 0, bucket_bytes: int = 64 << 20, group=None) -> int:
    """In-place broadcast of many tensors in few, large messages (per-link bound point-to-point xGMI: large messages, not
    thousands of small ones). A tensor of at least ``bucket_bytes / 4`` is broadcast IN PLACE (no staging copy: the big
    weight matrices never exist twice); smaller ones are coalesced into staging buckets of at most ``bucket_bytes``
    (the only transient memory). Returns the number of collectives issued."""
    import torch.distributed as dist

    calls = 0
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size, calls
        by_key = {}
        for t in bucket:
            by_key.setdefault((t.dtype, t.device), []).append(t)
        for ts in by_key.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=group)
            calls += 1
            off = 0
            for t in ts:
                t.copy_(flat[off: off + t.numel()].view_as(t))
                off += t.numel()
        bucket, size = [], 0

    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if nbytes * 4 >= bucket_bytes and t.is_contiguous():
            dist.broadcast(t, src=src, group=group)
            calls += 1
            continue
        if bucket and size + nbytes > bucket_bytes:
            flush()
        bucket.append(t)
        size += nbytes
    if bucket:
        flush()
    return calls


def broadcast_model_weights(model, src: int = 0, bucket_bytes: int = 64 << 20) -> int:
    """Weights of a ParlerTTSForConditionalGeneration (incl. the DAC wrapper's tensors) from rank ``src`` to all."""
    dev = model.device
    model.audio_encoder._weights = {k: v.to(dev) for k, v in model.audio_encoder._weights.items()}
    tensors = [p.data for p in model.parameters()] + list(model.audio_encoder._weights.values())
    n = broadcast_tensors(tensors, src=src, bucket_bytes=bucket_bytes)
    model._engine = None  # engines re-pack lazily from the received tensors
    model.audio_encoder._engine = None
    return n


def max_over_ranks(value: float, device=None) -> float:
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
