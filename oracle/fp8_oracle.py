"""TEST INFRASTRUCTURE ONLY — independent restatement of the weight-only e4m3 quantiser the product applies in its fp8-weight
mode (parler_tts_amd/quant.py uses torch's float8_e4m3fn cast; this file rounds by hand), and the helper that turns a decoder
state dict into the quantised model the oracle then evaluates with its ordinary bf16 arithmetic.

OCP FP8 E4M3 (FN variant: no infinities, max 448): 1 sign, 4 exponent (bias 7), 3 mantissa bits; normal range 2^-6 .. 448,
subnormals are multiples of 2^-9. Round to nearest, ties to even. Scales are powers of two per output row,
scale = 2^ceil(log2(max|w| / 448)), so q * scale is exact in bf16 (4 significant bits of q).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

E4M3_MAX = 448.0


def round_to_e4m3(x: torch.Tensor) -> torch.Tensor:
    """fp32 values (|x| <= 448) → nearest e4m3fn-representable fp32 values, ties to even."""
    x = x.double()
    mant, exp = torch.frexp(x.abs())  # |x| = mant * 2^exp, mant in [0.5, 1)
    e = exp - 1                        # |x| in [2^e, 2^(e+1))
    step = torch.where(e >= -6, torch.exp2((e - 3).double()), torch.full_like(x, 2.0 ** -9))  # 3 mantissa bits; subnormal grid 2^-9
    q = torch.round(x / step) * step   # torch.round: half to even
    return q.clamp(-E4M3_MAX, E4M3_MAX).float()


def quantize_rows(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """w [N, K] → (dequantised fp32 [N, K], scale [N])."""
    w = w.float()
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, torch.exp2(torch.ceil(torch.log2(amax / E4M3_MAX))), torch.ones_like(amax))
    return round_to_e4m3((w / scale[:, None]).clamp(-E4M3_MAX, E4M3_MAX)) * scale[:, None], scale


def quantize_kv_rows(x: torch.Tensor) -> torch.Tensor:
    """The e4m3 self-attention cache of the product's kv_fp8 mode (csrc/ptts_lm_kernels.h: kv8_row_scale / kv8_pack8), restated: every row of 64
    head dimensions (last axis) gets ONE power-of-two scale 2^ceil(log2(max|x| / 448)) - taken from the exponent of max|x| / 448, exactly as the
    kernel does (frexp: a power of two keeps its own exponent, anything above rounds up) - and its values are rounded to e4m3 (nearest, ties
    to even). Returns the dequantised rows (what attention sees for EVERY position, the newest included)."""
    x = x.float()
    amax = x.abs().amax(dim=-1, keepdim=True)
    m, e = torch.frexp(amax / E4M3_MAX)  # amax / 448 = m * 2^e, m in [0.5, 1)
    ce = torch.where(m == 0.5, e - 1, e)
    scale = torch.where(amax > 0, torch.exp2(ce.float()), torch.ones_like(amax))
    return round_to_e4m3(x / scale) * scale


FP8_MATRICES = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.out_proj", "encoder_attn.q_proj",
                "encoder_attn.out_proj", ".fc1.", ".fc2.")


def quantize_decoder_weights(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The model of the product's fp8-weight mode: every decode-step projection matrix (and the LM heads) replaced by its e4m3
    dequantisation; embedding tables, LayerNorms and the cross-attention K/V projections stay as they are."""
    out = {}
    for k, v in sd.items():
        is_mat = k.endswith(".weight") and ((k.startswith("lm_heads.")) or ("model.decoder.layers." in k and any(s in k for s in FP8_MATRICES)))
        out[k] = quantize_rows(v)[0] if is_mat else v
    return out
