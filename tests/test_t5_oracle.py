"""CPU: the T5 description-encoder oracle (oracle/t5_oracle.py) pinned against the installed transformers T5EncoderModel - the reference's own
text encoder class (modeling_parler_tts.py:2345-2348) - and the C ABI's host-side bucket function pinned against transformers' too."""
import numpy as np
import pytest
import torch

from oracle import t5_oracle as TO


def _cases():
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, 128, (3, 21), generator=g)
    mask = torch.ones(3, 21, dtype=torch.long)
    mask[1, 13:] = 0     # right padding
    mask[2, :4] = 0      # left padding (the reference pads descriptions on the left in batched inference)
    return ids, mask


@pytest.mark.parametrize("masked", [False, True])
def test_oracle_matches_transformers(masked):
    spec = TO.T5Spec()
    sd = TO.make_t5_weights(spec, seed=3)
    ids, mask = _cases()
    hf = TO.hf_encoder(spec, sd)
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=mask if masked else None).last_hidden_state
    out = TO.T5Oracle(spec, sd).encode(ids, mask if masked else None, zero_masked=False)
    assert float((out - ref).abs().max()) <= 1e-5


def test_oracle_long_description_and_fully_masked_row():
    """140 tokens: relative positions beyond max_distance (bucket saturation); one utterance with EVERY position masked (the additive
    finfo.min mask leaves a uniform softmax there, not NaN)."""
    spec = TO.T5Spec(num_layers=1)
    sd = TO.make_t5_weights(spec, seed=4)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 128, (2, 140), generator=g)
    mask = torch.ones(2, 140, dtype=torch.long)
    mask[1] = 0
    hf = TO.hf_encoder(spec, sd)
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=mask).last_hidden_state
    out = TO.T5Oracle(spec, sd).encode(ids, mask, zero_masked=False)
    assert torch.isfinite(out).all()
    assert float((out - ref).abs().max()) <= 1e-5


def test_bucket_restatements_match_transformers():
    from transformers.models.t5.modeling_t5 import T5Attention

    from parler_tts_amd import _native as N

    lib = N.load_library()
    rel = torch.arange(-1100, 1101)
    for nb, md in ((32, 128), (16, 64), (32, 256)):
        ref = T5Attention._relative_position_bucket(rel, bidirectional=True, num_buckets=nb, max_distance=md)
        assert torch.equal(TO.relative_position_bucket(rel, nb, md), ref)
        got = torch.tensor([lib.ptts_t5_relative_bucket(int(r), nb, md) for r in rel.tolist()])
        assert torch.equal(got, ref), (nb, md, rel[got != ref].tolist())


def test_golden_fixture_matches_oracle():
    """tests/golden/t5_tiny.npz = outputs of the installed transformers T5EncoderModel (oracle/make_golden_t5.py); the GPU tests compare the
    HIP encoder with the same file."""
    import os

    from conftest import GOLD

    z = np.load(os.path.join(GOLD, "t5_tiny.npz"))
    spec = TO.T5Spec(**{k: (float(v) if k == "layer_norm_epsilon" else int(v)) for k, v in zip(z["spec_keys"].tolist(), z["spec_vals"].tolist())})
    sd = TO.make_t5_weights(spec, seed=int(z["seed"]))
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    out = TO.T5Oracle(spec, sd).encode(ids, mask, zero_masked=False)
    assert float((out - torch.from_numpy(z["hf_masked"])).abs().max()) <= 1e-5
    out = TO.T5Oracle(spec, sd).encode(ids, None)
    assert float((out - torch.from_numpy(z["hf_unmasked"])).abs().max()) <= 1e-5


def test_bf16_oracle_is_a_small_perturbation_of_fp32():
    spec = TO.T5Spec()
    sd = TO.make_t5_weights(spec, seed=3)
    ids, mask = _cases()
    a = TO.T5Oracle(spec, sd).encode(ids, mask)
    b = TO.T5Oracle(spec, sd, precision="bf16").encode(ids, mask)
    rel = float((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt())
    assert 1e-5 < rel < 2e-2, rel
