"""CPU: the oracle restatements reproduce the golden vectors that oracle/make_golden.py generated from the
REFERENCE's own classes (decoder logits, greedy ids, delay pattern, EOS gate) and the DAC cross-check."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from helpers import spec_from_gold, t
from oracle import dac_oracle as DA
from oracle import decoder_oracle as DO


@pytest.mark.parametrize("variant", ["sin", "rope", "gqa"])
@pytest.mark.parametrize("attn", ["sdpa", "eager"])
def test_decoder_logits_match_reference(variant, attn):
    g = np.load(os.path.join(GOLD, f"decoder_{variant}.npz"))
    spec = spec_from_gold(g["spec"])
    sd = DO.make_decoder_weights(spec, seed=int(g["weight_seed"]))
    orc = DO.DecoderOracle(spec, sd, attn_impl=attn)
    K = spec.num_codebooks
    bsz = g["enc"].shape[0]
    ids0 = torch.full((bsz * K, 1), spec.bos_token_id, dtype=torch.long)
    out = orc.forward(ids0, t(g["enc"]), t(g["enc_mask"]), t(g["prompt"]), t(g["prompt_mask"]))
    assert (out[:, -1] - t(g["prefill_logits"])).abs().max() < 2e-6
    for s in range(g["step_ids"].shape[0]):
        out = orc.forward(t(g["step_ids"][s]))
        assert (out[:, -1] - t(g["step_logits"][s])).abs().max() < 2e-6


@pytest.mark.parametrize("variant", ["sin", "rope"])
def test_greedy_ids_match_reference_driven_loop(variant):
    g = np.load(os.path.join(GOLD, f"greedy_{variant}.npz"))
    spec = spec_from_gold(g["spec"])
    sd = DO.make_decoder_weights(spec, seed=int(g["weight_seed"]))
    for k in range(spec.num_codebooks):
        sd[f"lm_heads.{k}.weight"][spec.eos_token_id] *= float(g["eos_row_gain"])
    gp = DO.GenParams(max_length=int(g["max_length"]), min_new_tokens=int(g["min_new_tokens"]))
    tr = DO.sample_loop(DO.DecoderOracle(spec, sd), t(g["enc"]), t(g["enc_mask"]), t(g["prompt"]), t(g["prompt_mask"]), gp)
    assert torch.equal(tr.sequences, t(g["sequences"]))
    assert torch.equal(DO.undelay(tr.sequences, spec, gp.max_length), t(g["codes"]))
    assert tr.min_margin == pytest.approx(float(g["min_margin"]), rel=1e-3)


def test_delay_pattern_known_answers():
    g = np.load(os.path.join(GOLD, "delay_kat.npz"))
    for ci in range(int(g["n"])):
        K, seq_len, max_len, bsz = [int(x) for x in g[f"c{ci}_args"]]
        ids, mask = DO.build_delay_pattern_mask(t(g[f"c{ci}_in"]), 1025, 1024, max_len, K)
        assert torch.equal(ids, t(g[f"c{ci}_ids"])) and torch.equal(mask, t(g[f"c{ci}_mask"]))


def test_eos_gate_known_answers():
    g = np.load(os.path.join(GOLD, "eosgate_kat.npz"))
    K, bsz = int(g["K"]), int(g["bsz"])
    gate = DO.EosGate(1024, K, bsz)
    hist = t(g["history"])
    for s in range(g["gated"].shape[0]):
        sc = gate(hist[:, : s + 2], torch.zeros(bsz * K, 1088))
        assert np.array_equal(torch.isinf(sc[:, 1024]).numpy(), g["gated"][s])


def test_dac_restatement_matches_golden():
    g = np.load(os.path.join(GOLD, "dac_tiny.npz"))
    sd = DA.make_dac_weights(DA.DAC_TINY, seed=int(g["weight_seed"]), weight_norm_format="parametrized")
    orc = DA.DacOracle(DA.DAC_TINY, sd)
    assert (orc.from_codes(t(g["codes"])) - t(g["latents"])).abs().max() < 1e-6
    assert (orc.decode(t(g["codes"])) - t(g["wav"])).abs().max() < 1e-5


def test_dac_encode_restatement_matches_golden():
    """Voice-prompt path: latents to fp32 rounding, codes identical on every frame whose top-2 score gap is clear of it."""
    g = np.load(os.path.join(GOLD, "dac_tiny_encode.npz"))
    sd = DA.make_dac_weights(DA.DAC_TINY, seed=int(g["weight_seed"]), weight_norm_format="parametrized", with_encoder=True)
    orc = DA.DacOracle(DA.DAC_TINY, sd)
    wave = t(g["wave"])
    padded = orc.preprocess(wave)
    assert padded.shape[-1] % DA.DAC_TINY.hop_length == 0 and padded.shape[-1] - wave.shape[-1] < DA.DAC_TINY.hop_length
    z = orc.encode_latents(padded)
    assert (z - t(g["latents"])).abs().max() < 1e-5
    codes, margin = orc.quantize(z)
    safe = (t(g["margin"]) >= 1e-4)[:, None, :].expand_as(codes)
    assert torch.equal(codes[safe], t(g["codes"])[safe])
    assert torch.equal(orc.encode(wave), codes)
    assert orc.encode(wave, n_quantizers=3).shape[1] == 3 and torch.equal(orc.encode(wave, n_quantizers=3), codes[:, :3])


def test_weight_norm_formats_fold_identically():
    a = DA.fold_weight_norm(DA.make_dac_weights(DA.DAC_TINY, 7, "legacy"))
    b = DA.fold_weight_norm(DA.make_dac_weights(DA.DAC_TINY, 7, "parametrized"))
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k])
