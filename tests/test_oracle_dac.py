"""CPU: the DAC restatement (oracle/dac_oracle.py) against the only other on-disk statement of the same model, the independent
transformers ``DacModel`` port, LIVE on the same weights (descript-audio-codec itself is a third-party dependency that is not
installed: oracle header). Decode at the tiny test widths and at the real 44.1 kHz widths (1024 -> 1536,
strides 8, 8, 4, 2; 54 M parameters); encode (voice prompt) latents and codes at the tiny widths. transformers travels with the
image, so unlike the /root/reference pins this check also runs on the GPU box."""
import math

import pytest
import torch

from oracle import dac_oracle as DA
from oracle.make_golden import hf_dac_port

transformers = pytest.importorskip("transformers")
if not hasattr(transformers, "DacModel"):
    pytest.skip("this transformers has no DacModel", allow_module_level=True)


@pytest.mark.parametrize("name,T", [("tiny", 13), ("44k", 5)])
@torch.no_grad()
def test_decode_restatement_equals_transformers_port(name, T):
    spec = DA.DAC_TINY if name == "tiny" else DA.DAC_44KHZ
    sd = DA.make_dac_weights(spec, seed=4321, weight_norm_format="parametrized" if name == "tiny" else "folded")
    codes = torch.randint(0, spec.codebook_size, (2, spec.num_codebooks, T), generator=torch.Generator().manual_seed(3))
    wav = DA.DacOracle(spec, sd).decode(codes)
    port = hf_dac_port(spec, sd)
    wav2 = port.decode(audio_codes=codes).audio_values.reshape(wav.shape)
    assert wav.shape == (2, 1, T * spec.hop_length)
    rms = float(wav.pow(2).mean().sqrt())
    assert rms > 0.05 and float((wav - wav2).abs().max()) < 2e-5 * max(1.0, rms * 10)


@torch.no_grad()
def test_encode_restatement_equals_transformers_port():
    spec = DA.DAC_TINY
    sd = DA.make_dac_weights(spec, seed=4321, weight_norm_format="parametrized", with_encoder=True)
    g = torch.Generator().manual_seed(11)
    t = torch.arange(32 * 40 + 9) / 400.0
    wave = 0.4 * torch.sin(2 * math.pi * 3.0 * t)[None, None] * torch.tensor([1.0, 0.6])[:, None, None] + 0.2 * torch.randn(2, 1, t.numel(), generator=g)
    orc = DA.DacOracle(spec, sd)
    padded = orc.preprocess(wave)
    z = orc.encode_latents(padded)
    codes, margin = orc.quantize(z)
    port = hf_dac_port(spec, sd)
    assert float((z - port.encoder(padded)).abs().max()) < 1e-4 * float(z.abs().max())
    codes2 = port.encode(padded).audio_codes
    safe = margin >= 1e-4  # frames whose nearest-code search is clear of fp32 rounding at every stage
    assert int(safe.sum()) > safe.numel() // 2
    assert bool((codes == codes2)[safe[:, None, :].expand_as(codes)].all())
