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


@pytest.mark.parametrize("name,T", [("tiny", 13), ("44k", 5)])
@torch.no_grad()
def test_bf16_operand_oracle_equals_port_with_rounded_conv_operands(name, T):
    """DacOracle(precision="bf16") — the model the bf16-operand HIP kernels evaluate — stated a second, independent way: the transformers
    port with (a) the decoder's conv weights rounded to bf16 and (b) a forward pre-hook on every decoder conv EXCEPT the final Conv1d(C→1)
    that rounds its input to bf16 (biases / Snake / residual adds stay fp32 module arithmetic). Same rounding points ⇒ same waveform up to
    fp32 summation order (a value that lands on the other side of a bf16 boundary moves one operand by 2^-8: rare, bounded at 5e-4 rel. RMS);
    a rounding point put in the wrong place (e.g. the residual stream, or the final conv's input) shows up as ~1e-2."""
    spec = DA.DAC_TINY if name == "tiny" else DA.DAC_44KHZ
    sd = DA.make_dac_weights(spec, seed=4321)
    codes = torch.randint(0, spec.codebook_size, (2, spec.num_codebooks, T), generator=torch.Generator().manual_seed(3))
    wq = DA.DacOracle(spec, sd, precision="bf16").decode(codes)
    w32 = DA.DacOracle(spec, sd).decode(codes)
    port = hf_dac_port(spec, sd)
    convs = [m for m in port.decoder.modules() if isinstance(m, (torch.nn.Conv1d, torch.nn.ConvTranspose1d))]
    final = port.decoder.conv2
    assert final in convs and final.out_channels == 1
    for m in convs:
        if m is final:
            continue
        m.weight.data = DA._rb(m.weight.data)
        m.register_forward_pre_hook(lambda mod, args: (DA._rb(args[0]),) + tuple(args[1:]))
    wp = port.decode(audio_codes=codes).audio_values.reshape(wq.shape)
    rel = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())  # noqa: E731
    assert rel(wq, wp) <= 5e-4, rel(wq, wp)
    assert 2e-3 <= rel(wq, w32) <= 3e-2, rel(wq, w32)  # and it is a different model from the fp32 one (operand rounding through ~30 convs)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@torch.no_grad()
def test_stage_by_stage_restatement_chains_to_decode_and_negative_controls(prec):
    """DacOracle.stage (what tests/test_dac_stage_parity_gpu.py compares each HIP kernel with) chained over all stages IS decode_latents, bit for
    bit. Negative controls at the 44.1 kHz widths: what the per-stage bars (relative RMS of a unit's contribution to the stream <= 1.5e-4 in bf16 mode, measured 5e-5 on MI355X;
    bf16 flips of the inner activation under a 1e-6 perturbation cost ~6e-5 there) must and do catch -
    a residual unit whose inner activation is not rounded to bf16, a k7 conv with one tap dropped, an input slipped by one frame."""
    import torch.nn.functional as F

    spec = DA.DAC_44KHZ
    sd = DA.make_dac_weights(spec, seed=4321)
    o = DA.DacOracle(spec, sd, precision=prec)
    codes = torch.randint(0, 1024, (1, 9, 6), generator=torch.Generator().manual_seed(5))
    z = o.from_codes(codes)
    act, raw, keep = (DA._rb(z) if prec == "bf16" else z), None, {}
    for s in range(o.n_stages()):
        keep[s] = (act, raw)
        raw, act, _ = o.stage(s, act, raw)
    n = len(spec.decoder_rates)
    out = o.final_stage(act)  # the stage the GPU test pins conv_out_tanh_lds_kernel with (Conv1d(C -> 1, k7) + tanh on the last activation)
    assert torch.equal(out, torch.tanh(F.conv1d(act, o.w[f"decoder.model.{n + 2}.weight"], o.w[f"decoder.model.{n + 2}.bias"], padding=3)))
    assert torch.equal(out, o.decode(codes))
    # negative control for that stage's bar (relative RMS <= 1e-5, max |d| <= 1e-5 in tests/test_dac_stage_parity_gpu.py): an input slipped by one sample
    slipped = o.final_stage(torch.roll(act, 1, dims=-1))
    assert float((slipped - out).abs().max()) >= 1e-3
    if prec != "bf16":
        return
    rel = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())  # noqa: E731
    for s in (6, 10, 14, 15):  # residual units of the C = 384 / 192 / 96 blocks (the fused kernels)
        a_in, r_in = keep[s]
        good, _, y = o.stage(s, a_in, r_in)
        bi, k = divmod(s - 1, 4)
        r = f"decoder.model.{bi + 1}.block.{k + 1}.block."
        dil = (1, 3, 9)[k - 1]
        y32 = DA.snake1d(F.conv1d(a_in, o.w[r + "1.weight"], o.w[r + "1.bias"], dilation=dil, padding=3 * dil), o.w[r + "2.alpha"])  # NOT rounded
        unrounded = r_in + F.conv1d(y32, o.w[r + "3.weight"], o.w[r + "3.bias"])
        w7 = o.w[r + "1.weight"].clone()
        w7[:, :, 0] = 0  # a dropped tap
        yt = DA._rb(DA.snake1d(F.conv1d(a_in, w7, o.w[r + "1.bias"], dilation=dil, padding=3 * dil), o.w[r + "2.alpha"]))
        tap = r_in + F.conv1d(yt, o.w[r + "3.weight"], o.w[r + "3.bias"])
        slip, _, _ = o.stage(s, torch.roll(a_in, 1, dims=-1), r_in)  # the halo one frame off
        assert rel(unrounded - r_in, good - r_in) >= 8 * 1.5e-4, (s, rel(unrounded - r_in, good - r_in))  # measured 1.6e-3
        assert rel(tap - r_in, good - r_in) >= 1e-2 and rel(slip - r_in, good - r_in) >= 1e-2, (s, rel(tap, good), rel(slip, good))
