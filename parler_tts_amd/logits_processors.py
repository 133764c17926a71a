"""``ParlerTTSLogitsProcessor`` with the reference's constructor and call signature
(parler_tts/logits_processors.py:6-53). On the default ``generate()`` path the same gating runs inside the
device-side sampler tail (csrc/ptts_lm_kernels.h: tail_kernel); this host-side class exists for callers that pass
their own ``LogitsProcessorList`` (the `logits_processor=` argument replaces the default list, modeling:3418).
"""
from __future__ import annotations

import math

import torch


class ParlerTTSLogitsProcessor:
    """EOS may only be emitted by codebook k once codebook k-1 has emitted it (one step of delay per codebook)."""

    def __init__(self, eos_token_id, num_codebooks: int, batch_size: int, device: str = "cpu"):
        if not isinstance(eos_token_id, torch.Tensor):
            if isinstance(eos_token_id, int):
                eos_token_id = [eos_token_id]
            eos_token_id = torch.tensor(eos_token_id, device=device)
        if torch.is_floating_point(eos_token_id) or (eos_token_id < 0).any():
            raise ValueError(f"`eos_token_id` has to be a list of positive integers, but is {eos_token_id}")
        self.eos_token_id = eos_token_id
        self.batch_size = batch_size
        self.num_codebooks = num_codebooks
        self.device = device
        rows = batch_size * num_codebooks
        self.codebook_idx = torch.arange(rows, device=device)
        base = torch.arange(batch_size, device=device) * num_codebooks
        self.first_codebooks_unfinished = base.clone()
        self.max_codebooks = base + num_codebooks - 1

    def __call__(self, input_ids: torch.LongTensor, scores: torch.FloatTensor) -> torch.FloatTensor:
        eos = self.eos_token_id.to(input_ids.device)
        seen_eos = torch.isin(input_ids, eos).sum(1)
        cur = self.first_codebooks_unfinished
        advance = (seen_eos[cur] > 0) & (cur < self.max_codebooks)  # at most one codebook per step
        self.first_codebooks_unfinished = torch.where(advance, cur + 1, cur)
        blocked = self.codebook_idx > self.first_codebooks_unfinished.repeat_interleave(self.num_codebooks)
        rows = blocked.to(scores.device).nonzero(as_tuple=True)[0]
        for e in eos.tolist():
            scores[rows, e] = -math.inf
        return scores
