"""Seeded synthetic weights at real checkpoint shapes (no checkpoints are reachable offline): used by bench.py and
smoke runs. Variance-preserving init for the DAC stack (std = 1/sqrt(fan_in)) so the waveform is non-degenerate."""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch


def random_dac_state_dict(num_codebooks: int = 9, codebook_size: int = 1024, codebook_dim: int = 8, latent_dim: int = 1024,
                          decoder_dim: int = 1536, rates: Sequence[int] = (8, 8, 4, 2), seed: int = 4321) -> Dict[str, torch.Tensor]:
    """dac.model.DAC parameter names with torch's parametrized weight-norm keys (what DACModel checkpoints hold)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, k, transpose=False, gain=1.0):
        v = torch.randn(*((cin, cout, k) if transpose else (cout, cin, k)), generator=g) / math.sqrt(cin * k) * gain
        sd[name + ".parametrizations.weight.original0"] = v.flatten(1).norm(dim=1).view(-1, 1, 1)
        sd[name + ".parametrizations.weight.original1"] = v
        sd[name + ".bias"] = 0.01 * torch.randn(cout, generator=g)

    for i in range(num_codebooks):
        q = f"quantizer.quantizers.{i}."
        sd[q + "codebook.weight"] = torch.randn(codebook_size, codebook_dim, generator=g)
        conv(q + "out_proj", latent_dim, codebook_dim, 1, gain=1.0 / math.sqrt(num_codebooks))
    d = "decoder.model."
    conv(d + "0", decoder_dim, latent_dim, 7)
    for bi, s in enumerate(rates):
        cin, cout = decoder_dim // 2 ** bi, decoder_dim // 2 ** (bi + 1)
        b = f"{d}{bi + 1}.block."
        sd[b + "0.alpha"] = 0.5 + torch.rand(1, cin, 1, generator=g)
        conv(b + "1", cout, cin, 2 * s, transpose=True, gain=math.sqrt(s))
        for ri in range(3):
            r = f"{b}{ri + 2}.block."
            sd[r + "0.alpha"] = 0.5 + torch.rand(1, cout, 1, generator=g)
            conv(r + "1", cout, cout, 7, gain=0.5)
            sd[r + "2.alpha"] = 0.5 + torch.rand(1, cout, 1, generator=g)
            conv(r + "3", cout, cout, 1, gain=0.5)
    cl = decoder_dim // 2 ** len(rates)
    sd[f"{d}{len(rates) + 1}.alpha"] = 0.5 + torch.rand(1, cl, 1, generator=g)
    conv(f"{d}{len(rates) + 2}", 1, cl, 7)
    return sd
