"""CPU: the product's e4m3 weight quantiser (parler_tts_amd/quant.py, torch float8 cast) against the oracle's hand-rounded
restatement (oracle/fp8_oracle.py): identical dequantised values and scales, exact bf16 representability of the result."""
import torch

from oracle import fp8_oracle as FO
from parler_tts_amd import quant as Q


def test_quantiser_matches_hand_rounded_e4m3_and_is_exact_in_bf16():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(257, 512, generator=g) * 0.02
    w[0] = 0.0                                   # all-zero row: scale 1
    w[1, :8] = torch.tensor([448.0, -448.0, 447.9, 1e-9, 2.0 ** -9, 1.5 * 2.0 ** -9, 2.5 * 2.0 ** -9, 0.0])  # range ends, subnormal ties
    w[2] = torch.linspace(-3, 3, 512)            # every binade incl. ties
    w[3] *= 1e4
    q, scale, deq = Q.quantize_rows_e4m3(w)
    ref, rscale = FO.quantize_rows(w)
    assert q.dtype == torch.uint8 and q.shape == w.shape
    assert torch.equal(scale, rscale)
    assert torch.equal(deq, ref), float((deq - ref).abs().max())
    assert bool(((torch.log2(scale) % 1) == 0).all())                     # powers of two
    assert torch.equal(deq.to(torch.bfloat16).float(), deq)               # exact in bf16: one model for both weight formats
    assert float((deq / scale[:, None]).abs().max()) <= 448.0
    rel = ((deq - w).abs() / w.abs().clamp_min(1e-12))[w.abs() > scale[:, None] * 2.0 ** -6]
    assert float(rel.max()) <= 2.0 ** -4 + 1e-6                           # 3 mantissa bits: half an ulp = 2^-4 relative


def test_fp8_matrix_selection_matches_between_product_and_oracle():
    from oracle import decoder_oracle as DO

    sd = DO.make_decoder_weights(DO.TINY, seed=1)
    qsd = FO.quantize_decoder_weights(sd)
    for k, v in sd.items():
        assert Q.is_fp8_matrix(k) == (not torch.equal(qsd[k], v)), k
