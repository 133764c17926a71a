"""TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product path).

CPU restatement (plain torch, fp32) of the description encoder generate() runs once per call: transformers ``T5EncoderModel`` as the
reference builds it (``AutoModelForTextEncoding``, /root/reference/parler_tts/modeling_parler_tts.py:2345-2348; called at :3048-3097).
T5 is a THIRD-PARTY dependency of the reference (transformers, pinned 4.46.1 in /root/reference/setup.py; installed here: 5.15), so the
algorithm is restated from transformers' published ``modeling_t5.py`` and PINNED against the installed module itself:
tests/test_t5_oracle.py runs both on the same weights (fp32 <= 1e-5, incl. padding masks and a fully masked row) and
tests/golden/t5_tiny.npz holds outputs of the installed ``T5EncoderModel`` (oracle/make_golden_t5.py).

  T5Stack.forward            embed -> blocks -> final_layer_norm                                 (modeling_t5.py T5Stack)
  T5LayerSelfAttention       h + o(attn(layer_norm(h)))                                         (T5LayerSelfAttention.forward)
  T5Attention                scores = q k^T (no scaling) + position_bias + (1 - mask) * finfo.min; softmax fp32
  position bias              relative_attention_bias[bucket(key - query)], block 0's table for every block (T5Attention.compute_bias)
  T5LayerFF (gated-gelu)     h + wo(gelu_new(wi_0 x) * wi_1 x), x = layer_norm(h)                (T5DenseGatedActDense.forward)
  T5LayerNorm                x * rsqrt(mean(x^2) + eps) * weight                                 (T5LayerNorm.forward)

``precision="bf16"`` rounds where the HIP engine rounds (csrc/ptts_t5.hip): weights and every GEMM operand (normalised rows, attention
context, gated feed-forward activation) to bf16, everything else (accumulation, residual stream, attention, norms) in fp32. transformers'
own bf16 run rounds after EVERY op (residual stream, scores, probabilities): the engine is compared with this model, and both with fp32.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch


@dataclass
class T5Spec:
    vocab_size: int = 128
    d_model: int = 128
    d_kv: int = 64
    d_ff: int = 256
    num_layers: int = 2
    num_heads: int = 2
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6

    @property
    def inner(self) -> int:
        return self.num_heads * self.d_kv


FLAN_T5_LARGE = T5Spec(vocab_size=32128, d_model=1024, d_kv=64, d_ff=2816, num_layers=24, num_heads=16)


def make_t5_weights(spec: T5Spec, seed: int = 0, scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights under transformers' T5EncoderModel names. Variances follow T5PreTrainedModel._init_weights (factor 1.0):
    q ~ (d_model * d_kv)^-1/2, k / v / wi ~ d_model^-1/2, o ~ inner^-1/2, wo ~ d_ff^-1/2; norms near 1 (not exactly: a weight of all
    ones would hide a missing multiply)."""
    g = torch.Generator().manual_seed(seed)
    D, F, I = spec.d_model, spec.d_ff, spec.inner
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std * scale
    sd = {"shared.weight": rn(spec.vocab_size, D),
          "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": rn(spec.relative_attention_num_buckets, spec.num_heads, std=D ** -0.5 * 4),
          "encoder.final_layer_norm.weight": 1.0 + 0.1 * rn(D)}
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    for l in range(spec.num_layers):
        p = f"encoder.block.{l}."
        sd[p + "layer.0.SelfAttention.q.weight"] = rn(I, D, std=(D * spec.d_kv) ** -0.5)
        sd[p + "layer.0.SelfAttention.k.weight"] = rn(I, D, std=D ** -0.5)
        sd[p + "layer.0.SelfAttention.v.weight"] = rn(I, D, std=D ** -0.5)
        sd[p + "layer.0.SelfAttention.o.weight"] = rn(D, I, std=I ** -0.5)
        sd[p + "layer.0.layer_norm.weight"] = 1.0 + 0.1 * rn(D)
        sd[p + "layer.1.DenseReluDense.wi_0.weight"] = rn(F, D, std=D ** -0.5)
        sd[p + "layer.1.DenseReluDense.wi_1.weight"] = rn(F, D, std=D ** -0.5)
        sd[p + "layer.1.DenseReluDense.wo.weight"] = rn(D, F, std=F ** -0.5)
        sd[p + "layer.1.layer_norm.weight"] = 1.0 + 0.1 * rn(D)
    return sd


def relative_position_bucket(relative_position: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """T5Attention._relative_position_bucket(bidirectional=True), same expression order (fp32 log, truncation)."""
    nb = num_buckets // 2
    buckets = (relative_position > 0).to(torch.long) * nb
    rp = torch.abs(relative_position)
    max_exact = nb // 2
    is_small = rp < max_exact
    large = max_exact + (torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(is_small, rp, large)


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


class T5Oracle:
    def __init__(self, spec: T5Spec, sd: Dict[str, torch.Tensor], precision: str = "fp32", fold_norm: bool = False):
        """``fold_norm``: where the bf16 engine rounds at <= 256 rows (csrc/ptts_t5.hip): T5LayerNorm commutes with the projection, so every norm
        but block 0's first one feeds the GEMM the rounded g o x and scales the product by rstd afterwards (same algebra; in bf16 a
        different - equivalent - rounding point; in fp32 a re-association)."""
        assert precision in ("fp32", "bf16")
        self.spec, self.bf, self.fold = spec, precision == "bf16", fold_norm
        self.sd = {k: v.detach().float() for k, v in sd.items()}
        if "shared.weight" not in self.sd:
            self.sd["shared.weight"] = self.sd["encoder.embed_tokens.weight"]
        if self.bf:  # the engine holds every matrix (and the embedding) in bf16; norm weights and the bias table stay fp32
            for k, v in self.sd.items():
                if v.dim() == 2 and "relative_attention_bias" not in k:
                    self.sd[k] = v.bfloat16().float()

    def _r(self, x: torch.Tensor) -> torch.Tensor:  # a GEMM operand as the engine stores it
        return x.bfloat16().float() if self.bf else x

    def _norm(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        var = x.pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(var + self.spec.layer_norm_epsilon))

    def position_bias(self, n: int) -> torch.Tensor:
        """[heads, n, n]: bias[h, i, j] = table[bucket(j - i), h]."""
        pos = torch.arange(n)
        rel = pos[None, :] - pos[:, None]  # memory - context
        b = relative_position_bucket(rel, self.spec.relative_attention_num_buckets, self.spec.relative_attention_max_distance)
        return self.sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"][b].permute(2, 0, 1)

    @torch.no_grad()
    def encode(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, zero_masked: bool = True, upto: Optional[int] = None) -> torch.Tensor:
        """ids [B, N] -> last_hidden_state fp32 [B, N, d_model] (masked positions zeroed when ``zero_masked``, as generate() does,
        modeling_parler_tts.py:3093-3097). ``upto``: stop after that many blocks and return the residual stream (debugging)."""
        sp, sd = self.spec, self.sd
        B, N = input_ids.shape
        H, dk = sp.num_heads, sp.d_kv
        h = sd["shared.weight"][input_ids.long()]
        bias = self.position_bias(N)[None]  # [1, H, N, N]
        if attention_mask is not None:
            ext = (1.0 - attention_mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
            bias = bias + ext
        nl = sp.num_layers if upto is None else upto
        def normed_proj(hh, wname, mats, folded):
            """[x @ W.T for W in mats] with x = layer_norm(hh): the plain order, or the engine's folded one (operand g o h rounded, product * rstd)."""
            w = sd[wname]
            if not folded:
                x = self._r(self._norm(hh, w))
                return [x @ sd[m].T for m in mats]
            rstd = torch.rsqrt(hh.pow(2).mean(-1, keepdim=True) + sp.layer_norm_epsilon)
            x = self._r(hh * w)
            return [(x @ sd[m].T) * rstd for m in mats]

        for l in range(nl):
            p = f"encoder.block.{l}."
            q, k, v = [t.view(B, N, H, dk).transpose(1, 2) for t in normed_proj(
                h, p + "layer.0.layer_norm.weight", [p + f"layer.0.SelfAttention.{n}.weight" for n in "qkv"], self.fold and l > 0)]
            scores = q @ k.transpose(2, 3) + bias
            pr = torch.softmax(scores, dim=-1)
            ctx = self._r((pr @ v).transpose(1, 2).reshape(B, N, H * dk))
            h = h + ctx @ sd[p + "layer.0.SelfAttention.o.weight"].T
            u, g_ = normed_proj(h, p + "layer.1.layer_norm.weight", [p + "layer.1.DenseReluDense.wi_0.weight", p + "layer.1.DenseReluDense.wi_1.weight"], self.fold)
            ff = self._r(gelu_new(u) * g_)
            h = h + ff @ sd[p + "layer.1.DenseReluDense.wo.weight"].T
        if upto is not None:
            return h
        out = self._norm(h, sd["encoder.final_layer_norm.weight"])
        if attention_mask is not None and zero_masked:
            out = out * attention_mask[..., None].float()
        return out


def hf_config(spec: T5Spec):
    """The transformers T5Config of a spec (flan-t5 style: gated-gelu, untied)."""
    from transformers import T5Config

    return T5Config(vocab_size=spec.vocab_size, d_model=spec.d_model, d_kv=spec.d_kv, d_ff=spec.d_ff, num_layers=spec.num_layers,
                    num_heads=spec.num_heads, relative_attention_num_buckets=spec.relative_attention_num_buckets,
                    relative_attention_max_distance=spec.relative_attention_max_distance, layer_norm_epsilon=spec.layer_norm_epsilon,
                    feed_forward_proj="gated-gelu", tie_word_embeddings=False, dropout_rate=0.0)


def hf_encoder(spec: T5Spec, sd: Dict[str, torch.Tensor]):
    """The installed transformers T5EncoderModel carrying `sd` (the pin of this restatement; fixtures: oracle/make_golden_t5.py)."""
    from transformers import T5EncoderModel

    m = T5EncoderModel(hf_config(spec)).eval()
    missing, unexpected = m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=False)
    assert not unexpected and all("embed_tokens" in k or k == "shared.weight" for k in missing), (missing, unexpected)
    return m
