"""CPU: pins ``oracle.decoder_oracle.sample_loop`` (the restated `_sample` loop) against the `_sample`
of the transformers release installed here, driven through ``oracle/hf_sample_shim.py`` with the same model (the oracle's
forward), the same processors (transformers' MinNewTokens / warpers + the ParlerTTSLogitsProcessor — the reference's own
class when /root/reference is importable, else the pinned port) and the same RNG seed."""
import os
import sys

import pytest
import torch

from oracle import decoder_oracle as DO
from oracle import hf_sample_shim as HS


def _eos_gate_factory(spec):
    """ParlerTTSLogitsProcessor: the reference's own class (logits_processors.py:6-53) when the reference tree is present."""
    try:
        from oracle import reference_shims as RS

        RS.import_reference()
        from parler_tts.logits_processors import ParlerTTSLogitsProcessor as cls  # noqa: N813 — the reference's class

        src = "reference"
    except Exception:  # noqa: BLE001 — GPU box / no reference mounted
        from parler_tts_amd.logits_processors import ParlerTTSLogitsProcessor as cls  # noqa: N813

        src = "port"
    return (lambda bsz: cls(spec.eos_token_id, spec.num_codebooks, bsz, "cpu")), src


def _case(seed, bsz=2, eos_gain=6.0):
    spec = DO.TINY
    sd = DO.make_decoder_weights(spec, seed=1234 + seed)
    for k in range(spec.num_codebooks):
        sd[f"lm_heads.{k}.weight"][spec.eos_token_id] *= eos_gain
    g = torch.Generator().manual_seed(seed)
    N, P = 9, 4
    enc = torch.randn(bsz, N, spec.hidden_size, generator=g)
    prompt = torch.randn(bsz, P, spec.hidden_size, generator=g) * 0.5
    enc_mask = torch.ones(bsz, N, dtype=torch.long)
    enc_mask[-1, 6:] = 0
    prompt_mask = torch.ones(bsz, P, dtype=torch.long)
    prompt_mask[-1, :2] = 0
    enc = enc * enc_mask[..., None]
    return spec, sd, enc, enc_mask, prompt, prompt_mask


@pytest.mark.parametrize("seed", [0, 3])
def test_greedy_ids_identical_incl_eos_gate_min_new_tokens_padding_and_early_stop(seed):
    spec, sd, enc, enc_mask, prompt, prompt_mask = _case(seed)
    gp = DO.GenParams(max_length=41, min_new_tokens=10)
    factory, src = _eos_gate_factory(spec)
    with torch.no_grad():
        ref = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, enc_mask, prompt, prompt_mask, gp, keep_scores=True)
        seq, scores, names = HS.hf_sample(DO.DecoderOracle(spec, sd), enc, enc_mask, prompt, prompt_mask, gp, factory)
    # transformers' own order: built-ins (MinNewTokens) -> the custom list (EOS gate) -> [warpers]
    assert names == ["MinNewTokensLengthLogitsProcessor", "ParlerTTSLogitsProcessor"], names
    assert torch.equal(seq, ref.sequences), (src, seq.shape, ref.sequences.shape)
    assert (ref.sequences == spec.eos_token_id).any(), "case does not exercise EOS"
    assert len(scores) == len(ref.step_scores)
    for a, b in zip(scores, ref.step_scores):
        assert torch.equal(a, b)  # processed scores incl. the -inf pattern of MinNewTokens + EOS gate


def test_greedy_without_min_new_tokens_stops_early_and_pads_finished_rows():
    spec, sd, enc, enc_mask, prompt, prompt_mask = _case(5, eos_gain=40.0)
    gp = DO.GenParams(max_length=41, min_new_tokens=0)
    factory, _ = _eos_gate_factory(spec)
    with torch.no_grad():
        ref = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, enc_mask, prompt, prompt_mask, gp)
        seq, _, names = HS.hf_sample(DO.DecoderOracle(spec, sd), enc, enc_mask, prompt, prompt_mask, gp, factory)
    assert names == ["ParlerTTSLogitsProcessor"]
    assert torch.equal(seq, ref.sequences)
    assert seq.shape[1] < 41, "expected the EOS criterion to end the loop before max_length"


@pytest.mark.parametrize("temperature,top_k,top_p", [(0.7, 20, 0.9), (1.0, 50, 1.0), (1.3, 0, 0.8)])
def test_sampling_processed_scores_and_seeded_draws_identical(temperature, top_k, top_p):
    spec, sd, enc, enc_mask, prompt, prompt_mask = _case(1)
    gp = DO.GenParams(max_length=30, min_new_tokens=6, do_sample=True, temperature=temperature, top_k=top_k, top_p=top_p)
    factory, _ = _eos_gate_factory(spec)
    with torch.no_grad():
        torch.manual_seed(77)
        ref = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, enc_mask, prompt, prompt_mask, gp, keep_scores=True)
        torch.manual_seed(77)
        seq, scores, names = HS.hf_sample(DO.DecoderOracle(spec, sd), enc, enc_mask, prompt, prompt_mask, gp, factory)
    want = ["MinNewTokensLengthLogitsProcessor", "ParlerTTSLogitsProcessor"]
    want += ["TemperatureLogitsWarper"] if temperature != 1.0 else []
    want += ["TopKLogitsWarper"] if top_k else []
    want += ["TopPLogitsWarper"] if top_p < 1.0 else []
    assert names == want, names
    assert len(scores) == len(ref.step_scores)
    for s_, (a, b) in enumerate(zip(scores, ref.step_scores)):
        assert torch.equal(torch.isinf(a), torch.isinf(b)), s_       # identical support after top-k / top-p
        assert torch.allclose(a[~torch.isinf(a)], b[~torch.isinf(b)], rtol=0, atol=0), s_  # identical values
    assert torch.equal(seq, ref.sequences)  # same seed, same probabilities, same multinomial draws
