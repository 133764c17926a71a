"""Weight-only OCP e4m3 quantisation for the decode step (BASELINE configs[4]: "fp8 weights"): one POWER-OF-TWO scale per
output row, so that ``q * scale`` is exactly representable in bf16 (an e4m3 value has 4 significant bits, bf16 keeps 8):
the GEMV step streams the 1-byte weights, every other path (prefill, batch > 4) runs the ordinary bf16 kernels on the exact
dequantisation — one model, two storage formats."""
from __future__ import annotations

from typing import Tuple

import torch

E4M3_MAX = 448.0


def quantize_rows_e4m3(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """w [N, K] float → (q uint8 [N, K] e4m3fn bytes, scale float32 [N] powers of two, dequantised float32 [N, K] = q * scale)."""
    w32 = w.detach().float()
    amax = w32.abs().amax(dim=1)
    scale = torch.where(amax > 0, torch.exp2(torch.ceil(torch.log2(amax / E4M3_MAX))), torch.ones_like(amax))
    scaled = (w32 / scale[:, None]).clamp_(-E4M3_MAX, E4M3_MAX)
    try:
        q = scaled.to(torch.float8_e4m3fn)  # round-to-nearest-even
    except (RuntimeError, TypeError):  # a device build without fp8 conversion kernels: convert on the host
        q = scaled.cpu().to(torch.float8_e4m3fn).to(w.device)
    return q.view(torch.uint8), scale.contiguous(), q.float() * scale[:, None]


def is_fp8_matrix(name: str) -> bool:
    """Decoder state-dict names that get an e4m3 copy: the projection matrices the decode step streams (not the embedding
    tables, LayerNorms, or the cross-attention K/V projections, which only run at prefill)."""
    if name.startswith("lm_heads.") and name.endswith(".weight"):
        return True
    if not name.startswith("model.decoder.layers.") or not name.endswith(".weight"):
        return False
    return any(s in name for s in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj", "encoder_attn.q_proj",
                                   "encoder_attn.out_proj", ".fc1.", ".fc2."))
