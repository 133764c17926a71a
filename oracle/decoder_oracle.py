"""TEST INFRASTRUCTURE ONLY — CPU restatement (torch fp32) of the Parler-TTS generation hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline. The product path (``parler_tts_amd/``)
never imports it and fails loudly when the HIP library is missing.

Pinning status: the reference ships no tests or golden vectors, so the oracle is pinned
against outputs of the *reference's own classes* run in the build container
(``oracle/make_golden.py`` → ``tests/golden/*.npz``; re-checked live by
``tests/test_oracle_vs_reference.py`` whenever ``/root/reference`` is present):
  * decoder forward (prefill + cached steps, sinusoidal and RoPE, with/without padding masks) is
    pinned against ``ParlerTTSForCausalLM`` (modeling_parler_tts.py:1824) to <= 2e-6 absolute;
  * delay-pattern helpers and ``ParlerTTSLogitsProcessor`` are pinned bit-exactly against
    modeling_parler_tts.py:205-276 and logits_processors.py:6-53;
  * the sampling loop restates transformers==4.46.1 ``GenerationMixin._sample`` (setup.py:21 pins
    it; the package source is NOT under /root/reference and 4.46.1 is not installed). It is pinned
    against the transformers release that IS installed (5.x): ``oracle/hf_sample_shim.py`` lets the
    installed ``GenerationMixin._sample`` + ``_get_logits_processor`` + ``_get_stopping_criteria``
    drive this module's forward, and ``tests/test_sample_loop_vs_transformers.py`` requires identical
    token ids (greedy, EOS gate / min_new_tokens / finished-row padding / early stop) and identical
    processed scores + seeded draws under temperature / top-k / top-p. The per-step core of `_sample`
    and the processor order [built-ins incl. MinNewTokens] + [custom list] + [warpers] are unchanged
    between 4.46 and 5.x as far as the reference's call sites (modeling_parler_tts.py:3412-3572) can
    tell; a 4.46.1 wheel has never been available offline, so THAT exact version stays unverified.

All line numbers below are ``/root/reference/parler_tts/modeling_parler_tts.py`` unless noted.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# config
# --------------------------------------------------------------------------------------------
@dataclass
class DecoderSpec:
    """The integers of ParlerTTSDecoderConfig (configuration_parler_tts.py:111-172) the path needs."""

    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    ffn_dim: int = 4096
    num_codebooks: int = 9
    vocab_size: int = 1088
    max_position_embeddings: int = 4096
    rope_embeddings: bool = False
    rope_theta: float = 10000.0
    pad_token_id: int = 1024
    eos_token_id: int = 1024
    bos_token_id: int = 1025
    use_fused_lm_heads: bool = False
    num_key_value_heads: int = 0                  # 0 = num_attention_heads (MHA); configuration_parler_tts.py:152-158
    num_cross_attention_key_value_heads: int = 0  # 0 = num_key_value_heads

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def kv_heads(self) -> int:
        return self.num_key_value_heads or self.num_attention_heads

    @property
    def cross_kv_heads(self) -> int:
        return self.num_cross_attention_key_value_heads or self.kv_heads


MINI_V1 = DecoderSpec()  # helpers/model_init_scripts/init_model_600M.py:27-44
LARGE_V1 = DecoderSpec(hidden_size=1536, num_hidden_layers=30, num_attention_heads=24, ffn_dim=6144)  # init_large_model.py:25-43
TINY = DecoderSpec(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, ffn_dim=256, max_position_embeddings=256)


# --------------------------------------------------------------------------------------------
# synthetic weights with the reference's state-dict names and init (:1093-1102)
# --------------------------------------------------------------------------------------------
def sinusoidal_table(num_embeddings: int, embedding_dim: int) -> torch.Tensor:
    """ParlerTTSSinusoidalPositionalEmbedding.get_embedding, :346-359 (cos ‖ sin halves)."""
    half_dim = embedding_dim // 2
    emb = math.log(10000) / (half_dim - 1)
    emb = torch.exp(torch.arange(half_dim, dtype=torch.int64).float() * -emb)
    emb = torch.arange(num_embeddings, dtype=torch.int64).float().unsqueeze(1) * emb.unsqueeze(0)
    emb = torch.cat([torch.cos(emb), torch.sin(emb)], dim=1).view(num_embeddings, -1)
    if embedding_dim % 2 == 1:
        emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
    return emb.to(torch.float32)


def rope_tables(head_dim: int, theta: float, num_positions: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """ParlerTTSRotaryEmbedding.forward, :373-406: fp32 freqs = inv_freq ⊗ position, emb = cat(freqs, freqs)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    pos = torch.arange(num_positions, dtype=torch.int64).float()
    freqs = (inv_freq[:, None] @ pos[None, :]).transpose(0, 1)  # :401 (matmul form, fp32)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def make_decoder_weights(spec: DecoderSpec, seed: int = 1234, prefix: str = "") -> Dict[str, torch.Tensor]:
    """N(0, 0.02) Linear/Embedding, LayerNorm (1, 0): _init_weights :1093-1102, initializer_factor 0.02.

    LayerNorm affine parameters get a small seeded perturbation on top of (1, 0) so that a kernel that
    drops gamma/beta cannot pass parity (documented deviation from the reference init; the oracle and
    the HIP path see identical tensors).
    """
    g = torch.Generator().manual_seed(seed)
    H, Fd, K, V = spec.hidden_size, spec.ffn_dim, spec.num_codebooks, spec.vocab_size
    sd: Dict[str, torch.Tensor] = {}

    def normal(*shape):
        return torch.randn(*shape, generator=g) * 0.02

    def ln(name):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
        sd[name + ".bias"] = 0.05 * torch.randn(H, generator=g)

    p = prefix + "model.decoder."
    for k in range(K):
        sd[f"{p}embed_tokens.{k}.weight"] = normal(V + 1, H)  # :1353 vocab_size + 1 rows
    if not spec.rope_embeddings:
        sd[f"{p}embed_positions.weights"] = sinusoidal_table(spec.max_position_embeddings, H)
    for i in range(spec.num_hidden_layers):
        lp = f"{p}layers.{i}."
        for att, nkv in (("self_attn", spec.kv_heads), ("encoder_attn", spec.cross_kv_heads)):
            for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):
                rows = nkv * spec.head_dim if proj in ("k_proj", "v_proj") else H  # grouped-query attention: fewer K/V heads (:449-452)
                sd[f"{lp}{att}.{proj}.weight"] = normal(rows, H)  # bias=False :952
            ln(f"{lp}{att}_layer_norm")
        sd[f"{lp}fc1.weight"] = normal(Fd, H)
        sd[f"{lp}fc2.weight"] = normal(H, Fd)
        ln(f"{lp}final_layer_norm")
    ln(f"{p}layer_norm")
    if spec.use_fused_lm_heads:
        sd[f"{prefix}lm_heads.weight"] = normal(K * V, H)
    else:
        for k in range(K):
            sd[f"{prefix}lm_heads.{k}.weight"] = normal(V, H)
    return sd


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


# --------------------------------------------------------------------------------------------
# delay pattern (a1) — restates :205-211 and :214-276
# --------------------------------------------------------------------------------------------
def build_delay_pattern_mask(input_ids: torch.Tensor, bos_token_id: int, pad_token_id: int, max_length: int,
                             num_codebooks: int) -> Tuple[torch.Tensor, torch.Tensor]:
    input_ids = input_ids.reshape(-1, num_codebooks, input_ids.shape[-1])
    bsz, num_codebooks, seq_len = input_ids.shape
    shifted = torch.full((bsz, num_codebooks, max_length), -1, dtype=torch.long)
    if max_length < 2 * num_codebooks - 1:  # :246-247
        return input_ids.reshape(bsz * num_codebooks, -1), shifted.reshape(bsz * num_codebooks, -1)
    for k in range(num_codebooks):  # :250-252
        shifted[:, k, k: seq_len + k] = input_ids[:, k]
    j = torch.arange(max_length)[None, :]
    k = torch.arange(num_codebooks)[:, None]
    eos_pat = (j - k) >= (max_length - num_codebooks + 1)  # triu(diagonal=max_length-K+1) :256-258
    bos_pat = j <= k  # tril :260
    keep = ~(bos_pat | eos_pat)
    ids = keep * shifted + bos_pat * bos_token_id + eos_pat * pad_token_id  # :262-265
    first = ids[:, 0, :]
    start = (first == -1).nonzero()[:, 1]
    first_start_id = int(start.min()) if len(start) > 0 else seq_len  # :269-275
    pattern_mask = ids.reshape(bsz * num_codebooks, -1)
    return ids[..., :first_start_id].reshape(bsz * num_codebooks, -1), pattern_mask


def apply_delay_pattern_mask(input_ids: torch.Tensor, pattern_mask: torch.Tensor) -> torch.Tensor:
    seq_len = input_ids.shape[-1]
    m = pattern_mask[..., :seq_len]
    return torch.where(m == -1, input_ids, m)  # :210


# --------------------------------------------------------------------------------------------
# ParlerTTSLogitsProcessor (a13) — restates logits_processors.py:24-53
# --------------------------------------------------------------------------------------------
class EosGate:
    def __init__(self, eos_token_id: int, num_codebooks: int, batch_size: int):
        self.eos = eos_token_id
        self.K = num_codebooks
        self.codebook_idx = torch.arange(batch_size * num_codebooks)
        self.first_unfinished = torch.arange(batch_size) * num_codebooks
        self.max_codebooks = torch.arange(batch_size) * num_codebooks + num_codebooks - 1

    def __call__(self, input_ids: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        is_eos = (input_ids == self.eos).sum(1)  # isin(...).sum(1) :46
        fu = self.first_unfinished
        self.first_unfinished = torch.where((is_eos[fu] > 0) & (fu < self.max_codebooks), fu + 1, fu)  # :48
        gate = self.codebook_idx > self.first_unfinished.repeat_interleave(self.K)  # :51
        scores[gate, self.eos] = -math.inf  # :52
        return scores


# --------------------------------------------------------------------------------------------
# decoder forward (a4-a12)
# --------------------------------------------------------------------------------------------
def _rotate_half(x):  # :409-413
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class DecoderOracle:
    """Functional restatement of ParlerTTSForCausalLM.forward (:1865) at inference with a KV cache.

    precision:
      "fp32" — everything fp32 (the reference's CPU float32 path, BASELINE.json configs[0]).
      "bf16" — the engine's throughput numerics: weights and K/V caches rounded to bf16, Linear inputs
               rounded to bf16, fp32 accumulation and fp32 residual stream / LayerNorm / softmax. This is
               the same quantised model the HIP bf16 path evaluates; the two differ only by summation order.
    attn_impl: "sdpa" (default registry entry, :933-937) or "eager" (:494-584; q pre-scaled :514).
    """

    def __init__(self, spec: DecoderSpec, sd: Dict[str, torch.Tensor], precision: str = "fp32",
                 prefix: str = "", attn_impl: str = "sdpa"):
        self.spec, self.precision, self.attn_impl = spec, precision, attn_impl
        self.p = prefix + "model.decoder."
        self.hp = prefix
        wr = bf16_round if precision == "bf16" else (lambda t: t)
        self.w: Dict[str, torch.Tensor] = {}
        for k, v in sd.items():
            if not k.startswith(prefix):
                continue
            v = v.detach().to(torch.float32)
            is_matrix = k.endswith("_proj.weight") or ".fc1." in k or ".fc2." in k or "lm_heads" in k or "embed_tokens" in k
            self.w[k] = wr(v) if is_matrix else v
        if spec.rope_embeddings:
            self.cos, self.sin = rope_tables(spec.head_dim, spec.rope_theta, spec.max_position_embeddings)
        self.reset()

    # -- state ---------------------------------------------------------------------------------
    def reset(self):
        L = self.spec.num_hidden_layers
        self.k_self: List[Optional[torch.Tensor]] = [None] * L
        self.v_self: List[Optional[torch.Tensor]] = [None] * L
        self.k_cross: List[Optional[torch.Tensor]] = [None] * L
        self.v_cross: List[Optional[torch.Tensor]] = [None] * L
        self.past_len = 0
        self.enc_mask = None
        self.prompt_mask = None
        self.trace: Dict[str, torch.Tensor] = {}

    # -- helpers -------------------------------------------------------------------------------
    def _act(self, x):
        return bf16_round(x) if self.precision == "bf16" else x

    def _linear(self, x, name):
        return F.linear(self._act(x), self.w[name])

    def _ln(self, x, name):
        return F.layer_norm(x, (self.spec.hidden_size,), self.w[name + ".weight"], self.w[name + ".bias"], 1e-5)

    def _heads(self, t, bsz, q, n=None):
        return t.view(bsz, q, n or self.spec.num_attention_heads, self.spec.head_dim).transpose(1, 2)

    def _repeat_kv(self, t):
        """repeat_kv :280-289: [B, n_kv, L, d] -> [B, n_heads, L, d], query head h reads K/V head h // n_rep."""
        n_rep = self.spec.num_attention_heads // t.shape[1]
        if n_rep == 1:
            return t
        b, nkv, L, d = t.shape
        return t[:, :, None].expand(b, nkv, n_rep, L, d).reshape(b, nkv * n_rep, L, d)

    def _attend(self, q, k, v, add_mask, causal):
        """q [B,h,Q,d], k/v [B,h,L,d]; add_mask broadcastable additive [B,1,Q,L] or None."""
        d = q.shape[-1]
        if self.attn_impl == "eager":
            s = torch.matmul(q * d ** -0.5, k.transpose(2, 3))  # :514 pre-scale, :552
        else:
            s = torch.matmul(q, k.transpose(2, 3)) * d ** -0.5  # SDPA scales inside
        if causal:
            Q, L = q.shape[2], k.shape[2]
            i = torch.arange(Q)[:, None] + (L - Q)
            s = s.masked_fill(torch.arange(L)[None, :] > i, -math.inf)
        if add_mask is not None:
            s = s + add_mask
        return torch.matmul(torch.softmax(s, dim=-1), v)

    # -- forward -------------------------------------------------------------------------------
    def embed(self, ids: torch.Tensor) -> torch.Tensor:
        """ids [B·K, T] → Σ_k embed_tokens[k](ids[:,k]) :1433."""
        K = self.spec.num_codebooks
        x = ids.reshape(-1, K, ids.shape[-1])
        return sum(F.embedding(x[:, k], self.w[f"{self.p}embed_tokens.{k}.weight"]) for k in range(K))

    def forward(self, ids: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None,
                encoder_attention_mask: Optional[torch.Tensor] = None,
                prompt_hidden_states: Optional[torch.Tensor] = None,
                prompt_attention_mask: Optional[torch.Tensor] = None, trace: bool = False) -> torch.Tensor:
        """One call of the decoder: prefill when the cache is empty (prompt prepended :1437-1439),
        otherwise one cached step. Returns logits [B·K, T, V] for ALL T positions of this call."""
        spec = self.spec
        x = self.embed(ids)
        if prompt_hidden_states is not None and self.past_len == 0:
            x = torch.cat([prompt_hidden_states.to(torch.float32), x], dim=1)
        bsz, T, H = x.shape
        if self.past_len == 0:
            self.enc_mask = encoder_attention_mask
            self.prompt_mask = prompt_attention_mask
        pos = torch.arange(self.past_len, self.past_len + T)
        cos = sin = None
        if not spec.rope_embeddings:
            x = x + self.w[f"{self.p}embed_positions.weights"].index_select(0, pos)  # :1506-1511
        else:
            cos, sin = self.cos[pos][None, None], self.sin[pos][None, None]  # :1533-1534 (unsqueeze_dim=1 :433)
        total = self.past_len + T
        # self-attention padding mask: [prompt_attention_mask ‖ ones] :1474-1501 / :2944-2969 → additive :1710-1724
        self_mask = None
        if self.prompt_mask is not None:
            pm = self.prompt_mask.to(torch.float32)
            full = torch.cat([pm, torch.ones(bsz, total - pm.shape[1])], dim=1)
            self_mask = ((1.0 - full) * torch.finfo(torch.float32).min)[:, None, None, :]
        cross_mask = None
        if self.enc_mask is not None:
            em = self.enc_mask.to(torch.float32)
            cross_mask = ((1.0 - em) * torch.finfo(torch.float32).min)[:, None, None, :]  # :1553-1562
        if trace:
            self.trace["embed"] = x.clone()
        for i in range(spec.num_hidden_layers):
            lp = f"{self.p}layers.{i}."
            # --- self attention :1019-1034
            r = x
            hn = self._ln(x, lp + "self_attn_layer_norm")
            q = self._heads(self._linear(hn, lp + "self_attn.q_proj.weight"), bsz, T)
            k = self._heads(self._linear(hn, lp + "self_attn.k_proj.weight"), bsz, T, spec.kv_heads)
            v = self._heads(self._linear(hn, lp + "self_attn.v_proj.weight"), bsz, T, spec.kv_heads)
            if spec.rope_embeddings:
                q = q * cos + _rotate_half(q) * sin  # :858-859
                k = k * cos + _rotate_half(k) * sin  # :880-882
            if getattr(self, "kv_fp8", False):  # the product's opt-in e4m3 self-attention cache (ptts_config::kv_fp8): quantised from the fp32 rows
                from .fp8_oracle import quantize_kv_rows

                k, v = quantize_kv_rows(k), quantize_kv_rows(v)
            else:
                k, v = self._act(k), self._act(v)  # bf16 mode: KV cache holds bf16
            if self.k_self[i] is None:
                self.k_self[i], self.v_self[i] = k, v
            else:
                self.k_self[i] = torch.cat([self.k_self[i], k], dim=2)  # DynamicCache.update :887-889
                self.v_self[i] = torch.cat([self.v_self[i], v], dim=2)
            a = self._attend(q, self._repeat_kv(self.k_self[i]), self._repeat_kv(self.v_self[i]), self_mask, causal=T > 1)
            a = a.transpose(1, 2).reshape(bsz, T, H)
            x = r + self._linear(a, lp + "self_attn.out_proj.weight")
            if trace:
                self.trace[f"l{i}.self"] = x.clone()
            # --- cross attention :1036-1055
            r = x
            hn = self._ln(x, lp + "encoder_attn_layer_norm")
            q = self._heads(self._linear(hn, lp + "encoder_attn.q_proj.weight"), bsz, T)
            if spec.rope_embeddings:
                q = q * cos + _rotate_half(q) * sin  # quirk: q rotated (:858-859), keys are not (:880)
            if self.k_cross[i] is None:  # computed once, then reused (:872-875)
                e = encoder_hidden_states.to(torch.float32)
                N = e.shape[1]
                self.k_cross[i] = self._act(self._heads(self._linear(e, lp + "encoder_attn.k_proj.weight"), bsz, N, spec.cross_kv_heads))
                self.v_cross[i] = self._act(self._heads(self._linear(e, lp + "encoder_attn.v_proj.weight"), bsz, N, spec.cross_kv_heads))
            a = self._attend(q, self._repeat_kv(self.k_cross[i]), self._repeat_kv(self.v_cross[i]), cross_mask, causal=False)
            a = a.transpose(1, 2).reshape(bsz, T, H)
            x = r + self._linear(a, lp + "encoder_attn.out_proj.weight")
            if trace:
                self.trace[f"l{i}.cross"] = x.clone()
            # --- FFN :1057-1064 (exact erf GELU, configuration:118)
            r = x
            hn = self._ln(x, lp + "final_layer_norm")
            hn = F.gelu(self._linear(hn, lp + "fc1.weight"))
            x = r + self._linear(hn, lp + "fc2.weight")
            if trace:
                self.trace[f"l{i}.ffn"] = x.clone()
        x = self._ln(x, self.p + "layer_norm")  # :1632
        self.past_len = total
        K, V = spec.num_codebooks, spec.vocab_size
        if spec.use_fused_lm_heads:  # :1917-1918
            logits = self._linear(x, self.hp + "lm_heads.weight").view(bsz, T, K, V).transpose(1, 2)
        else:
            logits = torch.stack([self._linear(x, f"{self.hp}lm_heads.{k}.weight") for k in range(K)], dim=1)  # :1920
        return logits.reshape(bsz * K, T, V)  # :1960


# --------------------------------------------------------------------------------------------
# sampling loop (a2, a14) — transformers 4.46.1 GenerationMixin._sample restated, + generate() post
# --------------------------------------------------------------------------------------------
@dataclass
class GenParams:
    max_length: int  # total columns incl. the BOS column (= 1 + max_new_tokens)
    min_new_tokens: int = 0
    do_sample: bool = False
    temperature: float = 1.0
    top_k: int = 0
    top_p: float = 1.0
    use_eos_gate: bool = True  # default LogitsProcessorList([ParlerTTSLogitsProcessor]) :3418


@dataclass
class GenTrace:
    sequences: torch.Tensor  # raw ids [B·K, Lout]
    step_logits: List[torch.Tensor] = field(default_factory=list)  # fp32 [B·K, V] before processors
    step_scores: List[torch.Tensor] = field(default_factory=list)  # fp32 [B·K, V] after processors + warpers (what the selection sees)
    min_margin: float = float("inf")  # min top-2 margin over unfinished rows (greedy tie-safety)


def sample_loop(model: DecoderOracle, enc: torch.Tensor, enc_mask: Optional[torch.Tensor],
                prompt: Optional[torch.Tensor], prompt_mask: Optional[torch.Tensor], gp: GenParams,
                generator: Optional[torch.Generator] = None, keep_logits: bool = False,
                decoder_input_ids: Optional[torch.Tensor] = None, keep_scores: bool = False) -> GenTrace:
    """``decoder_input_ids`` [bsz*K, T]: un-delayed audio codes of a voice prompt (modeling:3136-3194); the BOS column is
    prepended (:3017-3018), the delay pattern built over the prefix (:3523-3530) and the first forward runs over all given
    columns at once, as the reference does."""
    spec = model.spec
    K = spec.num_codebooks
    bsz = enc.shape[0]
    eos, pad, bos = spec.eos_token_id, spec.pad_token_id, spec.bos_token_id
    model.reset()
    seq = torch.full((bsz * K, 1), bos, dtype=torch.long)  # :3011-3014
    if decoder_input_ids is not None and decoder_input_ids.shape[-1] > 0:
        seq = torch.cat([seq, decoder_input_ids.long()], dim=-1)
    seq, pattern = build_delay_pattern_mask(seq, bos, pad, gp.max_length, K)  # :3523-3530
    given = seq.shape[-1]
    gate = EosGate(eos, K, bsz) if gp.use_eos_gate else None
    unfinished = torch.ones(bsz * K, dtype=torch.long)
    tr = GenTrace(sequences=seq)
    step = 0
    while True:
        fed = apply_delay_pattern_mask(seq, pattern)  # :2909
        if step == 0:
            logits = model.forward(fed, enc, enc_mask, prompt, prompt_mask)
        else:
            logits = model.forward(fed[:, -1:])  # :2930 (only the new column; prompt dropped :2915-2917)
        scores = logits[:, -1, :].clone().float()
        if keep_logits:
            tr.step_logits.append(scores.clone())
        # processors: [MinNewTokensLength] + [ParlerTTSLogitsProcessor] + warpers (4.46.1 _get_logits_processor order)
        if gp.min_new_tokens > 0 and (seq.shape[-1] - given) < gp.min_new_tokens:
            scores[:, eos] = -math.inf
        if gate is not None:
            scores = gate(seq, scores)
        if gp.do_sample:
            if gp.temperature != 1.0:
                scores = scores / gp.temperature
            if gp.top_k and gp.top_k > 0:
                kth = torch.topk(scores, min(gp.top_k, scores.shape[-1]))[0][..., -1, None]
                scores = scores.masked_fill(scores < kth, -math.inf)
            if gp.top_p < 1.0:
                sl, si = torch.sort(scores, descending=False)
                cp = sl.softmax(dim=-1).cumsum(dim=-1)
                rm = cp <= (1 - gp.top_p)
                rm[..., -1:] = False
                scores = scores.masked_fill(rm.scatter(1, si, rm), -math.inf)
            if keep_scores:
                tr.step_scores.append(scores.clone())
            nxt = torch.multinomial(F.softmax(scores, dim=-1), 1, generator=generator).squeeze(1)
        else:
            if keep_scores:
                tr.step_scores.append(scores.clone())
            top2 = torch.topk(scores, 2, dim=-1)[0]
            m = (top2[:, 0] - top2[:, 1])[unfinished.bool()]
            if m.numel():
                tr.min_margin = min(tr.min_margin, float(m.min()))
            nxt = torch.argmax(scores, dim=-1)
        nxt = nxt * unfinished + pad * (1 - unfinished)
        seq = torch.cat([seq, nxt[:, None]], dim=-1)
        done = (nxt == eos) | (seq.shape[-1] >= gp.max_length)  # EosTokenCriteria | MaxLengthCriteria
        unfinished = unfinished & ~done.long()
        step += 1
        if unfinished.max() == 0:
            break
    tr.sequences = seq
    return tr


def teacher_forced_logits(model: DecoderOracle, enc: torch.Tensor, enc_mask: Optional[torch.Tensor], prompt: Optional[torch.Tensor],
                          prompt_mask: Optional[torch.Tensor], seq: torch.Tensor, max_length: int, n_pass: Optional[int] = None) -> List[torch.Tensor]:
    """Raw fp32 logits [B·K, V] of passes 0..n_pass-1 when the model is fed the GIVEN raw ids ``seq`` [B·K, >= n_pass] (delay
    pattern applied as in the loop, :2909): lets a test judge every choice of another implementation against this model's own
    scores at the same history, independently of earlier near-tie flips."""
    spec = model.spec
    K = spec.num_codebooks
    model.reset()
    _, pattern = build_delay_pattern_mask(seq[:, :1], spec.bos_token_id, spec.pad_token_id, max_length, K)
    n_pass = seq.shape[1] - 1 if n_pass is None else n_pass
    outs: List[torch.Tensor] = []
    for s in range(n_pass):
        fed = apply_delay_pattern_mask(seq[:, : s + 1], pattern)
        logits = model.forward(fed, enc, enc_mask, prompt, prompt_mask) if s == 0 else model.forward(fed[:, -1:])
        outs.append(logits[:, -1, :].float())
    return outs


def undelay(seq: torch.Tensor, spec: DecoderSpec, max_length: int, decoder_input_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """generate() post-processing :3585-3597 → codes [B, K, Lout-K] (may still contain ids >= codebook_size). With a voice
    prompt (``decoder_input_ids``) the pattern applied to the raw ids is the one built over the prefix (:3523-3530, :3586);
    the un-delay keep-mask is the BOS/PAD triangle pair of the BOS column (identical to the reference's mask built from the
    un-delayed `input_ids`, :3589-3596: audio codes are never BOS/PAD)."""
    K = spec.num_codebooks
    bos, pad = spec.bos_token_id, spec.pad_token_id
    bsz = seq.shape[0] // K
    first = seq[:, :1] if decoder_input_ids is None else torch.cat([seq[:, :1], decoder_input_ids.long()], dim=-1)
    _, pattern = build_delay_pattern_mask(first, bos, pad, max_length, K)
    out = apply_delay_pattern_mask(seq, pattern)
    _, m2 = build_delay_pattern_mask(seq[:, :1], bos, pad, out.shape[1], K)
    keep = (m2 != bos) & (m2 != pad)
    return out[keep].reshape(bsz, K, -1)


def valid_frames(codes_bk: torch.Tensor, codebook_size: int = 1024) -> torch.Tensor:
    """Per-sample column filter of the sequential branch :3627-3636: keep columns with no id >= codebook_size."""
    return codes_bk[:, (codes_bk >= codebook_size).sum(dim=0) == 0]
