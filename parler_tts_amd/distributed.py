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
    _check_same_layout(tensors, dev)
    n = broadcast_tensors(tensors, src=src, bucket_bytes=bucket_bytes)
    model._engine = None  # engines re-pack lazily from the received tensors
    model.audio_encoder._engine = None
    return n


def _check_same_layout(tensors, dev) -> None:
    """Every rank must walk the same tensor list (count and total size): a replica built differently (e.g. an un-tied embedding) would
    otherwise die inside a mismatched collective or hang. One tiny all-gather before the weights move."""
    import torch.distributed as dist

    world = dist.get_world_size()
    cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([len(tensors), sum(t.numel() * t.element_size() for t in tensors)], dtype=torch.int64, device=cdev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    layouts = {(int(e[0]), int(e[1])) for e in every}
    if len(layouts) != 1:
        raise RuntimeError(f"broadcast_model_weights: replicas differ in their tensor lists (count, bytes) = {sorted(layouts)}; "
                           "build every rank's model from the same config (ParlerTTSForConditionalGeneration(config, init_weights=False) on ranks != 0)")


def max_over_ranks(value: float, device=None) -> float:
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# kwargs of generate() whose leading dimension is the utterance batch (rows of `decoder_input_ids` are batch * num_codebooks)
_BATCH_KW = ("input_ids", "attention_mask", "prompt_input_ids", "prompt_attention_mask", "prompt_hidden_states", "input_values")


def _world(group=None):
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def generate_sharded(model, inputs=None, dst=0, group=None, **kwargs):
    """``model.generate(...)`` for a batch spread over the ranks of the process group (one process per GPU):
    the utterances are partitioned statically and contiguously (``shard_range``), every rank runs ITS utterances through its own
    replica, and the waveforms come back in input order - the launcher-side "copy device -> host per rank and concatenate" of
    §8(e), done with one small metadata all-gather and one padded gather. No collective touches the token loop. The reference's
    own pattern for collecting generated audio across ranks is pad-then-gather too (training/run_parler_tts_training.py:1156-1161,
    `accelerator.pad_across_processes` + `gather_for_metrics`).

    Every rank passes the FULL batch (same tensors on every rank, as a data loader that is not sharded yields them).
    ``dst``: rank (within ``group``) that receives the result, or ``None`` for all ranks. Returns (on the receiving ranks) a ``GenerateOutput`` with
    ``.sequences`` = waveform float32 [batch, samples] zero-padded to the longest utterance of the whole batch (on the CPU) and
    ``["audios_length"]`` = per-utterance lengths, exactly what the single-process call returns for the same batch under greedy
    decoding; ``None`` on the other ranks. With sampling each rank draws from its own torch generator, advanced by the index of the shard's first
    utterance (identical seeds on all ranks still give different shards; the draws are not those of the single-process call).
    Without an initialised process group this is ``model.generate`` (world size 1)."""
    import torch.distributed as dist

    from .modeling_parler_tts import GenerateOutput

    rank, world = _world(group)
    if rank < 0:
        raise ValueError("generate_sharded: this process is not a member of `group` (torch.distributed.get_rank(group) == -1)")
    K = model.config.decoder.num_codebooks
    full = dict(kwargs)
    if inputs is not None:
        full["input_ids"] = inputs
    enc_out = full.get("encoder_outputs")
    if enc_out is not None and not torch.is_tensor(enc_out):
        enc_out = enc_out[0] if isinstance(enc_out, (tuple, list)) else enc_out.last_hidden_state
    sizes = [int(full[k].shape[0]) for k in _BATCH_KW if full.get(k) is not None] + ([int(enc_out.shape[0])] if enc_out is not None else [])
    if not sizes:
        raise ValueError("generate_sharded needs a batched input (`input_ids` or `encoder_outputs`)")
    B = sizes[0]
    if any(s != B for s in sizes):
        raise ValueError(f"inconsistent batch sizes {sizes}")
    lo, hi = shard_range(B, rank, world)
    local = dict(full)
    for k in _BATCH_KW:
        if local.get(k) is not None:
            local[k] = local[k][lo:hi]
    if enc_out is not None:
        local["encoder_outputs"] = (enc_out[lo:hi],)
    if local.get("decoder_input_ids") is not None:
        local["decoder_input_ids"] = local["decoder_input_ids"][lo * K: hi * K]
    local.pop("streamer", None)
    if full.get("streamer") is not None:
        raise ValueError("generate_sharded: a streamer serves ONE utterance on one rank; call model.generate(streamer=...) there")
    local["return_dict_in_generate"] = True
    if hi > lo:
        if local.get("do_sample") or (local.get("do_sample") is None and getattr(getattr(model, "generation_config", None), "do_sample", False)):
            # sampling: generate() seeds its device sampler from torch's generator. A launcher that calls torch.manual_seed(s) identically on
            # every rank would give every shard the SAME stream; advance the generator by the shard's first utterance index so shards differ
            # (still deterministic for a given seed and world size; not the single-process stream - stated in the docstring)
            torch.randint(0, 2 ** 31, (lo + 1,))
        out = model.generate(**local)
        wav = out.sequences.detach().float().cpu()
        lens = [int(x) for x in out["audios_length"]]
    else:  # more ranks than utterances: this rank idles
        wav, lens = torch.zeros(0, 1), []
    if not (dist.is_available() and dist.is_initialized()):  # no process group: the plain call (a one-rank group still runs the collectives)
        return GenerateOutput(sequences=wav, audios_length=lens)
    # ---- metadata: rows and padded width of every rank's block ------------------------------------------------------------
    backend = dist.get_backend(group)
    cdev = model.device if backend == "nccl" else torch.device("cpu")  # RCCL moves device tensors, gloo host tensors
    # rows of the largest block: taken from what generate() RETURNED on every rank (one tiny max-all-reduce), not recomputed from the arguments -
    # a `generation_config=` object carrying num_return_sequences, or any other expansion of the batch, would otherwise leave one rank with more rows
    # than the metadata tensor holds: a shape error on that rank only, and the others hanging in the all_gather
    per_t = torch.tensor([len(lens)], dtype=torch.int64, device=cdev)
    dist.all_reduce(per_t, op=dist.ReduceOp.MAX, group=group)
    per = int(per_t.item())
    meta = torch.zeros(2 + per, dtype=torch.int64)
    meta[0], meta[1] = wav.shape[0], wav.shape[1]
    if lens:
        meta[2: 2 + len(lens)] = torch.tensor(lens, dtype=torch.int64)  # per-utterance sample counts travel as int64 (exact at any length)
    meta = meta.to(cdev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    rows = [int(m[0]) for m in metas]
    width = max([int(m[1]) for m in metas if int(m[0]) > 0] or [1])
    max_rows = max(rows)
    # ---- payload: one padded block of waveforms per rank (an idle rank contributes an all-zero block of the common shape) --------------
    block = torch.zeros(max_rows, width, dtype=torch.float32)
    if wav.shape[0]:
        block[: wav.shape[0], : wav.shape[1]] = wav
    block = block.to(cdev)
    if dst is None:
        blocks = [torch.zeros_like(block) for _ in range(world)]
        dist.all_gather(blocks, block, group=group)
    else:
        blocks = [torch.zeros_like(block) for _ in range(world)] if rank == dst else None
        # `dst` counts within `group` (like `rank` above); the collective addresses ranks of the default group
        dist.gather(block, blocks, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if rank != dst:
            return None
    parts, all_lens = [], []
    for r in range(world):
        parts.append(blocks[r][: rows[r]].cpu())
        all_lens += [int(v) for v in metas[r][2: 2 + rows[r]].tolist()]
    wav_all = torch.cat(parts, dim=0)
    longest = max(all_lens) if all_lens else 0
    return GenerateOutput(sequences=wav_all[:, : max(longest, 1)], audios_length=all_lens)
