"""TEST INFRASTRUCTURE ONLY — CPU restatement (torch fp32) of DAC ``decode`` (codes → waveform) and ``encode``
(waveform → codes, the voice-prompt path: dac_wrapper/modeling_dac.py:33-104 → ``model.preprocess`` :64, ``model.encode`` :95).

Reference call sites: parler_tts/dac_wrapper/modeling_dac.py:138 (``quantizer.from_codes``) and :139
(``model.decode``). The arithmetic lives in the third-party package ``descript-audio-codec``
(setup.py:24, UNPINNED, not vendored under /root/reference, not installed, no network), so this file
restates its published 44 kHz architecture: RVQ ``from_codes`` = Σ_i out_proj_i(codebook_i[codes_i])
(codebook_dim 8, 1×1 conv to 1024), decoder = Conv1d(1024→1536,k7) → 4 × [Snake, ConvTranspose1d(k=2s,
stride s, pad ⌈s/2⌉), 3 × ResidualUnit(dil 1,3,9: Snake, Conv k7 dilated, Snake, Conv k1, +skip)] with
strides (8,8,4,2) → Snake → Conv1d(96→1,k7) → tanh; Snake(x) = x + (α+1e-9)⁻¹·sin²(αx); every conv is
weight-normalised (w = g·v/‖v‖ per output channel; modeling_dac.py:148-164 re-applies it), folded here
once at load.

Encode: ``preprocess`` right-pads with zeros to a multiple of the hop; encoder = Conv1d(1→64,k7) → 4 × [3 × ResidualUnit
(dil 1,3,9) at dim/2, Snake, Conv1d(dim/2→dim, k=2s, stride s, pad ⌈s/2⌉)] with strides (2,4,8,8) → Snake → Conv1d(1024→latent,
k3); RVQ: per stage e = in_proj(residual) (1×1 conv to codebook_dim 8), nearest code by cosine similarity of the
L2-normalised e and codebook (ViT-VQGAN trick), z_q = out_proj(e + (codebook[idx] − e)) (the straight-through expression is
kept: it is not bit-identical to codebook[idx]), residual −= z_q.

Pinning status: "parity unpinned" against descript-audio-codec itself (absent). The restatement is
cross-checked against the only other on-disk statement of the same model — the independent
transformers-5.15 ``DacModel`` port (tests/test_oracle_dac.py) — and frozen into tests/golden/dac_*.npz.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass
class DacSpec:
    """dac.model.DAC(...) as constructed at dac_wrapper/modeling_dac.py:24-28 (+ descript 44 kHz defaults)."""

    num_codebooks: int = 9
    codebook_size: int = 1024
    codebook_dim: int = 8
    latent_dim: int = 1024
    decoder_dim: int = 1536
    decoder_rates: Tuple[int, ...] = (8, 8, 4, 2)
    sampling_rate: int = 44100
    frame_rate: int = 86
    encoder_dim: int = 64  # descript 44 kHz default; encoder strides are the decoder's reversed

    @property
    def hop_length(self) -> int:
        return int(math.prod(self.decoder_rates))

    @property
    def encoder_rates(self) -> Tuple[int, ...]:
        return tuple(reversed(self.decoder_rates))


DAC_44KHZ = DacSpec()
DAC_TINY = DacSpec(num_codebooks=9, latent_dim=64, decoder_dim=256, decoder_rates=(4, 2, 2, 2), encoder_dim=16)


def make_dac_weights(spec: DacSpec, seed: int = 4321, weight_norm_format: str = "folded", with_encoder: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded synthetic DAC weights under descript's module names (``quantizer.quantizers.i.*``,
    ``decoder.model.N.*``). Variance-preserving init (std = 1/sqrt(fan_in)) instead of N(0,0.02): with 0.02
    the 30-conv stack underflows to ~1e-12 and a waveform RMS tolerance would be vacuous.

    weight_norm_format: "folded" → ``.weight``; "legacy" → ``.weight_g/.weight_v``;
    "parametrized" → ``.parametrizations.weight.original0/original1`` (modeling_dac.py:148-157).
    """
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, k, transpose=False, gain=1.0):
        shape = (cin, cout, k) if transpose else (cout, cin, k)
        fan_in = cin * k / (1 if not transpose else 1)
        v = torch.randn(*shape, generator=g) / math.sqrt(fan_in) * gain
        bias = 0.01 * torch.randn(cout, generator=g)
        if weight_norm_format == "folded":
            sd[name + ".weight"] = v
        else:
            # weight_norm dim=0: one g per index of dim 0 (for ConvTranspose1d that is the INPUT channel)
            norm = v.flatten(1).norm(dim=1).view(-1, 1, 1)
            gg = norm * (1.0 + 0.1 * torch.randn(norm.shape, generator=g))
            if weight_norm_format == "legacy":
                sd[name + ".weight_g"], sd[name + ".weight_v"] = gg, v
            else:
                sd[name + ".parametrizations.weight.original0"] = gg
                sd[name + ".parametrizations.weight.original1"] = v
        sd[name + ".bias"] = bias

    def snake(name, c):
        sd[name + ".alpha"] = (0.5 + torch.rand(1, c, 1, generator=g))

    for i in range(spec.num_codebooks):
        q = f"quantizer.quantizers.{i}."
        sd[q + "codebook.weight"] = torch.randn(spec.codebook_size, spec.codebook_dim, generator=g)
        conv(q + "out_proj", spec.latent_dim, spec.codebook_dim, 1, gain=1.0 / math.sqrt(spec.num_codebooks))
    d = "decoder.model."
    ch = spec.decoder_dim
    conv(d + "0", ch, spec.latent_dim, 7)
    for bi, s in enumerate(spec.decoder_rates):
        cin, cout = ch // 2 ** bi, ch // 2 ** (bi + 1)
        b = f"{d}{bi + 1}.block."
        snake(b + "0", cin)
        conv(b + "1", cout, cin, 2 * s, transpose=True, gain=math.sqrt(s))  # each output sees k/s = 2 taps
        for ri in range(3):
            r = f"{b}{ri + 2}.block."
            snake(r + "0", cout)
            conv(r + "1", cout, cout, 7, gain=0.5)
            snake(r + "2", cout)
            conv(r + "3", cout, cout, 1, gain=0.5)
    cl = ch // 2 ** len(spec.decoder_rates)
    snake(f"{d}{len(spec.decoder_rates) + 1}", cl)
    conv(f"{d}{len(spec.decoder_rates) + 2}", 1, cl, 7)
    if with_encoder:  # drawn AFTER everything above: the decode-only fixtures keep their values
        for i in range(spec.num_codebooks):
            conv(f"quantizer.quantizers.{i}.in_proj", spec.codebook_dim, spec.latent_dim, 1)
        e = "encoder.block."
        dim = spec.encoder_dim
        conv(e + "0", dim, 1, 7)
        for bi, st in enumerate(spec.encoder_rates):
            b = f"{e}{bi + 1}.block."
            for ri in range(3):
                r = f"{b}{ri}.block."
                snake(r + "0", dim)
                conv(r + "1", dim, dim, 7, gain=0.5)
                snake(r + "2", dim)
                conv(r + "3", dim, dim, 1, gain=0.5)
            snake(b + "3", dim)
            conv(b + "4", 2 * dim, dim, 2 * st)
            dim *= 2
        n = len(spec.encoder_rates)
        snake(f"{e}{n + 1}", dim)
        conv(f"{e}{n + 2}", spec.latent_dim, dim, 3, gain=4.0)
    return sd


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """w = g · v / ‖v‖ over all dims but 0 (torch weight_norm default dim=0), for either key format."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.endswith(".weight_g") or k.endswith(".parametrizations.weight.original0"):
            base = k[: -len(".weight_g")] if k.endswith(".weight_g") else k[: -len(".parametrizations.weight.original0")]
            vv = sd.get(base + ".weight_v", sd.get(base + ".parametrizations.weight.original1"))
            norm = vv.flatten(1).norm(dim=1).view(-1, *([1] * (vv.dim() - 1)))
            out[base + ".weight"] = v * vv / norm
        elif k.endswith(".weight_v") or k.endswith(".parametrizations.weight.original1"):
            continue
        else:
            out[k] = v
    return out


def snake1d(x: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def _rb(x: torch.Tensor) -> torch.Tensor:
    """fp32 → bf16 (round-to-nearest-even) → fp32: the value a bf16 MFMA operand carries."""
    return x.to(torch.bfloat16).to(torch.float32)


class DacOracle:
    """precision="fp32": the restatement as written. precision="bf16": the model the bf16-OPERAND engine evaluates (what
    ``model.to(device, dtype=torch.bfloat16)`` selects, INFERENCE.md:29-32 → dac_wrapper/modeling_dac.py:138-139 on a bf16 codec):
    every MFMA conv reads bf16 weights and bf16 activations, accumulates / adds bias / adds the residual / evaluates Snake in fp32.
    Rounding points, each placed where the HIP kernels round (csrc/ptts_dac.hip):
      * conv weights of the decoder stack → bf16 once at load (pack_conv_bf16_kernel); biases, alphas, the RVQ tables and the final
        Conv1d(C→1) + tanh stay fp32;
      * the latent z = Σ_i out_proj_i(codebook_i[codes]) → bf16 (rvq_gather_kernel, z_bf16);
      * every Snake output that feeds a conv → bf16 (the producer epilogue's ``out_act``), INCLUDING the activation between the k7 and
        the k1 conv of a residual unit (the fused unit's LDS tile; the two-launch path rounds the same value);
      * the residual stream (transposed-conv output, unit outputs) is fp32 and never rounded; the last unit's Snake output, which feeds
        the fp32 final conv, is not rounded (``act_f32``).
    ``sin`` is exact here (the kernels use v_sin_f32 in this mode: their distance to this oracle includes that)."""

    def __init__(self, spec: DacSpec, sd: Dict[str, torch.Tensor], prefix: str = "", precision: str = "fp32"):
        assert precision in ("fp32", "bf16"), precision
        self.spec = spec
        self.precision = precision
        sd = {k[len(prefix):]: v.detach().to(torch.float32) for k, v in sd.items() if k.startswith(prefix)}
        self.w = fold_weight_norm(sd)
        if precision == "bf16":
            n = len(spec.decoder_rates)
            for k in list(self.w):
                if k.startswith("decoder.model.") and k.endswith(".weight") and not k.startswith(f"decoder.model.{n + 2}."):
                    self.w[k] = _rb(self.w[k])

    def from_codes(self, codes: torch.Tensor) -> torch.Tensor:
        """codes [B, K, T] int64 → z [B, latent, T] (ResidualVectorQuantize.from_codes)."""
        z = 0.0
        for i in range(codes.shape[1]):
            q = f"quantizer.quantizers.{i}."
            zp = F.embedding(codes[:, i, :], self.w[q + "codebook.weight"]).transpose(1, 2)
            z = z + F.conv1d(zp, self.w[q + "out_proj.weight"], self.w[q + "out_proj.bias"])
        return z

    def decode_latents(self, z: torch.Tensor) -> torch.Tensor:
        w, d = self.w, "decoder.model."
        rb = _rb if self.precision == "bf16" else (lambda v: v)  # operand rounding (identity in fp32)
        x = F.conv1d(rb(z), w[d + "0.weight"], w[d + "0.bias"], padding=3)
        for bi, s in enumerate(self.spec.decoder_rates):
            b = f"{d}{bi + 1}.block."
            x = rb(snake1d(x, w[b + "0.alpha"]))
            x = F.conv_transpose1d(x, w[b + "1.weight"], w[b + "1.bias"], stride=s, padding=math.ceil(s / 2))  # fp32 residual stream
            for ri, dil in enumerate((1, 3, 9)):
                r = f"{b}{ri + 2}.block."
                y = rb(snake1d(x, w[r + "0.alpha"]))
                y = F.conv1d(y, w[r + "1.weight"], w[r + "1.bias"], dilation=dil, padding=3 * dil)
                y = rb(snake1d(y, w[r + "2.alpha"]))
                y = F.conv1d(y, w[r + "3.weight"], w[r + "3.bias"])
                x = x + y
        n = len(self.spec.decoder_rates)
        x = snake1d(x, w[f"{d}{n + 1}.alpha"])  # feeds the fp32 final conv: not rounded
        x = F.conv1d(x, w[f"{d}{n + 2}.weight"], w[f"{d}{n + 2}.bias"], padding=3)
        return torch.tanh(x)

    # ---- one stage at a time (per-kernel parity: tests/test_dac_stage_parity_gpu.py) -------------------------------------------------
    def n_stages(self) -> int:
        return 1 + 4 * len(self.spec.decoder_rates)

    def stage(self, s: int, act_in: torch.Tensor, raw_in):
        """Stage `s` of decode_latents on GIVEN inputs (0 = decoder.model.0 on z; per block: the transposed conv, then residual unit 1, 2, 3):
        (act_in [B, C, T] = the activation the stage's first conv reads - already Snake'd and, in bf16 mode, bf16-valued -, raw_in = the
        residual stream before the stage) → (raw_out = the stream after the stage, act_out = Snake of it with the NEXT layer's alpha, rounded
        like decode_latents rounds it, y = the unit's inner activation or None). Same operations, same order as decode_latents."""
        w, d = self.w, "decoder.model."
        rb = _rb if self.precision == "bf16" else (lambda v: v)
        n = len(self.spec.decoder_rates)
        if s == 0:
            raw = F.conv1d(act_in, w[d + "0.weight"], w[d + "0.bias"], padding=3)
            return raw, rb(snake1d(raw, w[d + "1.block.0.alpha"])), None
        bi, k = divmod(s - 1, 4)
        b = f"{d}{bi + 1}.block."
        if k == 0:
            st = self.spec.decoder_rates[bi]
            raw = F.conv_transpose1d(act_in, w[b + "1.weight"], w[b + "1.bias"], stride=st, padding=math.ceil(st / 2))
            return raw, rb(snake1d(raw, w[b + "2.block.0.alpha"])), None
        ri, dil = k - 1, (1, 3, 9)[k - 1]
        r = f"{b}{ri + 2}.block."
        y = F.conv1d(act_in, w[r + "1.weight"], w[r + "1.bias"], dilation=dil, padding=3 * dil)
        y = rb(snake1d(y, w[r + "2.alpha"]))
        raw = raw_in + F.conv1d(y, w[r + "3.weight"], w[r + "3.bias"])
        if ri < 2:
            return raw, rb(snake1d(raw, w[f"{b}{ri + 3}.block.0.alpha"])), y
        if bi + 1 < n:
            return raw, rb(snake1d(raw, w[f"{d}{bi + 2}.block.0.alpha"])), y
        return raw, snake1d(raw, w[f"{d}{n + 1}.alpha"]), y  # feeds the fp32 final conv: not rounded

    def final_stage(self, act_in: torch.Tensor) -> torch.Tensor:
        """The last step of decode_latents on a GIVEN input: Conv1d(C → 1, k7, pad 3) + tanh of the fp32 activation the last residual unit
        hands over (stage n_stages() - 1's act_out). fp32 in both precisions, like decode_latents."""
        w, d, n = self.w, "decoder.model.", len(self.spec.decoder_rates)
        return torch.tanh(F.conv1d(act_in, w[f"{d}{n + 2}.weight"], w[f"{d}{n + 2}.bias"], padding=3))

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes [B, K, T] → waveform [B, 1, hop·T]  (DACModel.decode, modeling_dac.py:138-139)."""
        return self.decode_latents(self.from_codes(codes))

    # ---- encode (voice prompt) ------------------------------------------------------------------------------------
    def preprocess(self, wave: torch.Tensor) -> torch.Tensor:
        """Right-pad to a multiple of the hop (descript ``DAC.preprocess``; modeling_dac.py:64)."""
        L = wave.shape[-1]
        hop = self.spec.hop_length
        return F.pad(wave, (0, math.ceil(L / hop) * hop - L))

    def encode_latents(self, wave: torch.Tensor) -> torch.Tensor:
        """wave [B, 1, L] (L a multiple of the hop) → z [B, latent, L / hop]."""
        w, e = self.w, "encoder.block."
        x = F.conv1d(wave, w[e + "0.weight"], w[e + "0.bias"], padding=3)
        for bi, st in enumerate(self.spec.encoder_rates):
            b = f"{e}{bi + 1}.block."
            for ri, dil in enumerate((1, 3, 9)):
                r = f"{b}{ri}.block."
                y = snake1d(x, w[r + "0.alpha"])
                y = F.conv1d(y, w[r + "1.weight"], w[r + "1.bias"], dilation=dil, padding=3 * dil)
                y = snake1d(y, w[r + "2.alpha"])
                y = F.conv1d(y, w[r + "3.weight"], w[r + "3.bias"])
                x = x + y
            x = snake1d(x, w[b + "3.alpha"])
            x = F.conv1d(x, w[b + "4.weight"], w[b + "4.bias"], stride=st, padding=math.ceil(st / 2))
        n = len(self.spec.encoder_rates)
        x = snake1d(x, w[f"{e}{n + 1}.alpha"])
        return F.conv1d(x, w[f"{e}{n + 2}.weight"], w[f"{e}{n + 2}.bias"], padding=1)

    def quantize(self, z: torch.Tensor, n_quantizers: int | None = None):
        """Residual VQ encode: z [B, latent, T] → (codes [B, nq, T] int64, min top-2 score margin per (b, t) over stages)."""
        nq = self.spec.num_codebooks if n_quantizers is None else min(n_quantizers, self.spec.num_codebooks)
        residual = z
        codes, margins = [], []
        for i in range(nq):
            q = f"quantizer.quantizers.{i}."
            cb = self.w[q + "codebook.weight"]
            p = F.conv1d(residual, self.w[q + "in_proj.weight"], self.w[q + "in_proj.bias"])  # [B, cd, T]
            B, cd, T = p.shape
            enc = F.normalize(p.permute(0, 2, 1).reshape(B * T, cd))
            cbn = F.normalize(cb)
            dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cbn.t() + cbn.pow(2).sum(1, keepdim=True).t()
            top2 = (-dist).topk(2, dim=1)
            idx = (-dist).max(1)[1].reshape(B, T)
            margins.append((top2.values[:, 0] - top2.values[:, 1]).reshape(B, T))
            zq = F.embedding(idx, cb).transpose(1, 2)
            zq = p + (zq - p)  # straight-through expression, evaluated as written
            zq = F.conv1d(zq, self.w[q + "out_proj.weight"], self.w[q + "out_proj.bias"])
            residual = residual - zq
            codes.append(idx)
        return torch.stack(codes, dim=1), torch.stack(margins, 0).min(0).values

    def encode(self, wave: torch.Tensor, n_quantizers: int | None = None) -> torch.Tensor:
        """wave [B, 1, L] → codes [B, K, ceil(L / hop)] (DACModel.encode with one chunk, modeling_dac.py:88-99)."""
        return self.quantize(self.encode_latents(self.preprocess(wave)), n_quantizers)[0]
