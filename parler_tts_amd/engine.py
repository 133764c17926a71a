"""Thin Python owners of the two native engines (libptts_hip.so). torch is used only for device memory,
streams and dtype plumbing: every FLOP of the hot path runs in the HIP library. No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import _native as N
from .quant import is_fp8_matrix


def _stream_ptr(stream: Optional["torch.cuda.Stream"] = None, device=None) -> C.c_void_p:
    """Handle of the CURRENT stream of `device` (the engine's device, not torch's current device: a stream belongs to
    the device it was created on)."""
    s = stream if stream is not None else torch.cuda.current_stream(device)
    return C.c_void_p(s.cuda_stream)


def _shape_arr(t: torch.Tensor):
    return (C.c_int64 * t.dim())(*t.shape)


def rope_tables(head_dim: int, theta: float, num_positions: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp32 cos/sin exactly as ParlerTTSRotaryEmbedding.forward builds them (modeling_parler_tts.py:373-406):
    inv_freq = theta^(-2i/d), freqs = inv_freq ⊗ position (fp32 matmul), emb = cat(freqs, freqs)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    pos = torch.arange(num_positions, dtype=torch.int64).float()
    freqs = (inv_freq[:, None] @ pos[None, :]).transpose(0, 1)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().contiguous(), emb.sin().contiguous()


class DecoderEngine:
    """Owner of a ``ptts_engine``: the autoregressive ParlerTTSDecoder step + sampler on MI355X."""

    def __init__(self, *, hidden_size: int, num_layers: int, num_heads: int, ffn_dim: int, num_codebooks: int,
                 vocab_size: int, max_positions: int, rope: bool = False, rope_theta: float = 10000.0,
                 pad_token_id: int = 1024, eos_token_id: int = 1024, bos_token_id: int = 1025,
                 dtype: torch.dtype = torch.bfloat16, max_batch: int = 1, max_ctx: int = 2700, max_enc: int = 256,
                 max_prompt: int = 128, device: Optional[torch.device] = None, num_kv_heads: int = 0, num_cross_kv_heads: int = 0,
                 weights_fp8: bool = False, kv_fp8: bool = False):
        if not torch.cuda.is_available():
            raise N.NativeLibraryError("DecoderEngine needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = N.load_library()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"engine dtype must be float32 or bfloat16, got {dtype}")
        self.dtype = dtype
        self.K, self.V, self.H = num_codebooks, vocab_size, hidden_size
        self.rope, self.rope_theta, self.max_positions = rope, rope_theta, max_positions
        self.cfg = N.PttsConfig(hidden_size, num_layers, num_heads, ffn_dim, num_codebooks, vocab_size, max_positions, int(rope),
                                float(rope_theta), pad_token_id, eos_token_id, bos_token_id,
                                N.PTTS_BF16 if dtype == torch.bfloat16 else N.PTTS_F32, max_batch, max_ctx, max_enc, max_prompt,
                                self.device.index or 0, int(num_kv_heads or 0), int(num_cross_kv_heads or 0), int(bool(weights_fp8)), int(bool(kv_fp8)))
        self.weights_fp8 = bool(weights_fp8)
        self.kv_fp8 = bool(kv_fp8)
        if self.weights_fp8 and dtype != torch.bfloat16:
            raise ValueError("weights_fp8 needs the bfloat16 engine (e4m3 weights, bf16 activations)")
        self._h = C.c_void_p()
        N.check(self.lib.ptts_engine_create(C.byref(self.cfg), C.byref(self._h)), "ptts_engine_create")
        self.B = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.ptts_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- weights ---------------------------------------------------------------------------------------
    def load_weight(self, name: str, tensor: torch.Tensor):
        t = tensor.detach()
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.float()
        t = t.to(self.device).contiguous()
        dt = N.PTTS_BF16 if t.dtype == torch.bfloat16 else N.PTTS_F32
        N.check(self.lib.ptts_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), dt, _shape_arr(t), t.dim(), _stream_ptr(device=self.device)),
                f"ptts_load_weight({name})")
        # the engine re-packs asynchronously on the current stream; keep `t` alive until then
        torch.cuda.current_stream(self.device).synchronize()

    def load_weight_fp8(self, name: str, tensor: torch.Tensor):
        """weights_fp8 engines: quantise one projection matrix (``quant.quantize_rows_e4m3``), hand the exact dequantisation to
        the ordinary loader (MFMA-packed bf16 copy for prefill / batch > 4) and the e4m3 bytes + row scales to the GEMV step."""
        from .quant import quantize_rows_e4m3

        q, scale, deq = quantize_rows_e4m3(tensor.detach().to(self.device))
        self.load_weight(name, deq)
        q, scale = q.contiguous(), scale.contiguous()
        N.check(self.lib.ptts_load_weight_fp8(self._h, name.encode(), C.c_void_p(q.data_ptr()), C.c_void_p(scale.data_ptr()), _shape_arr(q), 2,
                                              _stream_ptr(device=self.device)), f"ptts_load_weight_fp8({name})")
        torch.cuda.current_stream(self.device).synchronize()

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        """Accepts the reference's decoder state-dict names, optionally under `prefix`
        (e.g. ``"decoder."`` for a ParlerTTSForConditionalGeneration checkpoint)."""
        for k, v in sd.items():
            if not k.startswith(prefix):
                continue
            name = k[len(prefix):]
            if name.startswith("model.decoder.") or name.startswith("lm_heads."):
                if "rotary_emb" in name:
                    continue
                if self.weights_fp8 and name == "lm_heads.weight":  # fused heads (:1834-1840): quantise per codebook slice
                    for k in range(self.K):
                        self.load_weight_fp8(f"lm_heads.{k}.weight", v[k * self.V:(k + 1) * self.V])
                elif self.weights_fp8 and is_fp8_matrix(name):
                    self.load_weight_fp8(name, v)
                else:
                    self.load_weight(name, v)
        if self.rope:
            cos, sin = rope_tables(self.H // self.cfg.num_heads, self.rope_theta, max(self.max_positions, self.cfg.max_ctx))  # RoPE has no position limit (:373-406)
            self.load_weight("rope_cos", cos)
            self.load_weight("rope_sin", sin)
        N.check(self.lib.ptts_weights_ready(self._h), "ptts_weights_ready")

    # -- generation -------------------------------------------------------------------------------------
    def set_gen_params(self, *, max_length: int, min_new_tokens: int = 0, do_sample: bool = False, temperature: float = 1.0,
                       top_k: int = 0, top_p: float = 1.0, use_eos_gate: bool = True, seed: int = 0):
        gp = N.PttsGenParams(int(max_length), int(min_new_tokens), int(bool(do_sample)), float(temperature), int(top_k or 0), float(top_p),
                             int(bool(use_eos_gate)), int(seed) & (2 ** 64 - 1))
        N.check(self.lib.ptts_set_gen_params(self._h, C.byref(gp)), "ptts_set_gen_params")
        self.max_length = int(max_length)

    def prefill(self, enc: torch.Tensor, enc_mask: Optional[torch.Tensor], prompt: Optional[torch.Tensor],
                prompt_mask: Optional[torch.Tensor], sample: bool = True):
        enc = enc.to(self.device, torch.float32).contiguous()
        B, Nn, H = enc.shape
        if H != self.H:
            raise ValueError(f"encoder_hidden_states width {H} != decoder hidden_size {self.H}")
        P = 0
        keep = [enc]
        pm = em = pr = None
        if prompt is not None and prompt.shape[1] > 0:
            pr = prompt.to(self.device, torch.float32).contiguous()
            P = pr.shape[1]
            keep.append(pr)
        if enc_mask is not None:
            em = enc_mask.to(self.device, torch.int32).contiguous()
            keep.append(em)
        if prompt_mask is not None and P > 0:
            pm = prompt_mask.to(self.device, torch.int32).contiguous()
            keep.append(pm)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()
        N.check(self.lib.ptts_prefill(self._h, ptr(enc), ptr(em), ptr(pr), ptr(pm), B, Nn, P, int(sample), _stream_ptr(device=self.device)), "ptts_prefill")
        self.B, self.P = B, P
        self._keep = keep  # inputs are consumed asynchronously by the enqueued kernels

    def first_token_sync(self):
        """Blocks until the first token of the last sampling ``prefill`` exists on the device (``ptts_first_token_sync``): the end of
        time-to-first-token; work the prefill enqueued behind the sampler tail (cross-attention fold) is not waited for."""
        N.check(self.lib.ptts_first_token_sync(self._h), "ptts_first_token_sync")

    def first_token_times(self) -> Tuple[float, float]:
        """(prefill_ms, tail_ms): GPU time of the last sampling ``prefill`` as it ran in sequence on its stream - staging + forward up to
        the sampler tail, and the tail itself (``ptts_first_token_times``; synchronises on the first-token event)."""
        pre, tail = C.c_float(), C.c_float()
        N.check(self.lib.ptts_first_token_times(self._h, C.byref(pre), C.byref(tail)), "ptts_first_token_times")
        return float(pre.value), float(tail.value)

    def graph_nodes(self) -> int:
        """Kernel nodes per decode step of the step graph captured last (``ptts_debug_graph_nodes``)."""
        n = C.c_int32()
        N.check(self.lib.ptts_debug_graph_nodes(self._h, C.byref(n)), "ptts_debug_graph_nodes")
        return int(n.value)

    def set_audio_prefix(self, codes: Optional[torch.Tensor]):
        """Voice prompt for the NEXT ``prefill``: un-delayed audio codes int64 [B, K, T] (or [B*K, T]); ``None`` clears it."""
        if codes is None or codes.shape[-1] == 0:
            N.check(self.lib.ptts_set_audio_prefix(self._h, C.c_void_p(), 1, 0, _stream_ptr(device=self.device)), "ptts_set_audio_prefix")
            return
        codes = codes.to(self.device, torch.int64).reshape(-1, codes.shape[-1]).contiguous()
        if codes.shape[0] % self.K:
            raise ValueError(f"audio prefix rows {codes.shape[0]} not a multiple of num_codebooks {self.K}")
        N.check(self.lib.ptts_set_audio_prefix(self._h, C.c_void_p(codes.data_ptr()), codes.shape[0] // self.K, codes.shape[1], _stream_ptr(device=self.device)),
                "ptts_set_audio_prefix")
        self._keep_prefix = codes

    def decode_steps(self, n: int):
        N.check(self.lib.ptts_decode_steps(self._h, int(n), _stream_ptr(device=self.device)), "ptts_decode_steps")

    def state(self) -> Tuple[int, bool]:
        cur, fin = C.c_int32(), C.c_int32()
        N.check(self.lib.ptts_state(self._h, C.byref(cur), C.byref(fin), _stream_ptr(device=self.device)), "ptts_state")
        return cur.value, bool(fin.value)

    def _copy_out(self, ptr: int, shape, dtype, row_stride: Optional[int] = None) -> torch.Tensor:
        """Device-to-device copy of engine-owned memory into a fresh torch tensor (async on the current stream)."""
        out = torch.empty(tuple(shape), dtype=dtype, device=self.device)
        esz = out.element_size()
        hip = N.hip_runtime()
        if row_stride is None or row_stride == shape[-1]:
            rc = hip.hipMemcpyAsync(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(out.numel() * esz), 3, _stream_ptr(device=self.device))
        else:
            rc = hip.hipMemcpy2DAsync(C.c_void_p(out.data_ptr()), C.c_size_t(shape[-1] * esz), C.c_void_p(ptr), C.c_size_t(row_stride * esz),
                                      C.c_size_t(shape[-1] * esz), C.c_size_t(shape[0]), 3, _stream_ptr(device=self.device))
        if rc != 0:
            raise N.NativeLibraryError(f"hipMemcpy(D2D) failed with code {rc}")
        return out

    def ids(self, max_cols: Optional[int] = None) -> torch.Tensor:
        """Raw ids [B*K, cur_len] (a copy), exactly what `_sample` returns before the delay mask is applied. ``max_cols`` caps the
        columns returned: a caller reading while later steps are still in flight on another stream passes the number of columns it
        KNOWS to be complete (the device-side length may already count a column whose ids are not visible yet)."""
        p, ld = C.c_void_p(), C.c_int32()
        N.check(self.lib.ptts_ids(self._h, C.byref(p), C.byref(ld)), "ptts_ids")
        cur, _ = self.state()
        if max_cols is not None:
            cur = min(cur, int(max_cols))
        return self._copy_out(p.value, (self.B * self.K, cur), torch.int64, row_stride=ld.value)

    def step_forward(self):
        N.check(self.lib.ptts_step_forward(self._h, _stream_ptr(device=self.device)), "ptts_step_forward")

    def logits(self) -> torch.Tensor:
        """fp32 [B*K, V] logits of the last forward (a copy)."""
        p = C.c_void_p()
        N.check(self.lib.ptts_logits(self._h, C.byref(p)), "ptts_logits")
        return self._copy_out(p.value, (self.B * self.K, self.V), torch.float32)

    def push_tokens(self, tokens: torch.Tensor, finished: Optional[torch.Tensor] = None):
        tk = tokens.to(self.device, torch.int64).contiguous()
        fn = finished.to(self.device, torch.int32).contiguous() if finished is not None else None
        N.check(self.lib.ptts_push_tokens(self._h, C.c_void_p(tk.data_ptr()), C.c_void_p(fn.data_ptr()) if fn is not None else C.c_void_p(),
                                          _stream_ptr(device=self.device)), "ptts_push_tokens")
        self._keep2 = (tk, fn)

    def generate_ids(self, enc, enc_mask, prompt, prompt_mask, poll_every: int = 64, audio_prefix: Optional[torch.Tensor] = None) -> torch.Tensor:
        """prefill + graph-replayed decode until every row finished; returns raw ids [B*K, Lout]. ``audio_prefix``:
        un-delayed voice-prompt codes [B, K, T] continued by the decoder."""
        self.set_audio_prefix(audio_prefix)
        self.prefill(enc, enc_mask, prompt, prompt_mask, sample=True)
        remaining = self.max_length - 2 - (0 if audio_prefix is None else int(audio_prefix.shape[-1]))
        while remaining > 0:
            n = min(poll_every, remaining)
            self.decode_steps(n)
            remaining -= n
            _, fin = self.state()
            if fin:
                break
        return self.ids()


class T5Engine:
    """Owner of a ``ptts_t5``: the description encoder (transformers ``T5EncoderModel``, gated-gelu, d_kv 64) on MI355X.
    Replaces the ``self.text_encoder(...)`` call of generate() (modeling_parler_tts.py:3048-3097) on the time-to-first-token path."""

    def __init__(self, *, vocab_size: int, d_model: int, d_kv: int, d_ff: int, num_layers: int, num_heads: int,
                 relative_attention_num_buckets: int = 32, relative_attention_max_distance: int = 128, layer_norm_epsilon: float = 1e-6,
                 dtype: torch.dtype = torch.bfloat16, max_batch: int = 1, max_len: int = 64, device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise N.NativeLibraryError("T5Engine needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = N.load_library()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"engine dtype must be float32 or bfloat16, got {dtype}")
        self.dtype, self.d_model, self.max_batch, self.max_len = dtype, d_model, max_batch, max_len
        self.cfg = N.PttsT5Config(vocab_size, d_model, d_kv, d_ff, num_layers, num_heads, relative_attention_num_buckets,
                                  relative_attention_max_distance, float(layer_norm_epsilon),
                                  N.PTTS_BF16 if dtype == torch.bfloat16 else N.PTTS_F32, max_batch, max_len, self.device.index or 0)
        self._h = C.c_void_p()
        N.check(self.lib.ptts_t5_create(C.byref(self.cfg), C.byref(self._h)), "ptts_t5_create")

    @staticmethod
    def supports(config) -> bool:
        """True for the T5 configurations the HIP encoder implements (flan-t5 / T5 v1.1 shapes: gated gelu_new feed-forward, d_kv 64)."""
        return (getattr(config, "model_type", None) == "t5" and getattr(config, "d_kv", 0) == 64 and bool(getattr(config, "is_gated_act", False))
                and getattr(config, "dense_act_fn", None) == "gelu_new" and config.d_model % 32 == 0 and config.d_ff % 32 == 0)

    @classmethod
    def from_config(cls, config, **kw) -> "T5Engine":
        return cls(vocab_size=config.vocab_size, d_model=config.d_model, d_kv=config.d_kv, d_ff=config.d_ff, num_layers=config.num_layers,
                   num_heads=config.num_heads, relative_attention_num_buckets=config.relative_attention_num_buckets,
                   relative_attention_max_distance=getattr(config, "relative_attention_max_distance", 128),
                   layer_norm_epsilon=config.layer_norm_epsilon, **kw)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.ptts_t5_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        """``sd``: transformers ``T5EncoderModel`` names (``shared.weight``, ``encoder.block.N....``), optionally under ``prefix``
        (``"text_encoder."`` for a ParlerTTSForConditionalGeneration checkpoint)."""
        for k, v in sd.items():
            if not k.startswith(prefix):
                continue
            name = k[len(prefix):]
            if not (name.startswith("encoder.") or name == "shared.weight"):
                continue
            t = v.detach()
            if t.dtype not in (torch.float32, torch.bfloat16):
                t = t.float()
            t = t.to(self.device).contiguous()
            dt = N.PTTS_BF16 if t.dtype == torch.bfloat16 else N.PTTS_F32
            N.check(self.lib.ptts_t5_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), dt, _shape_arr(t), t.dim(), _stream_ptr(device=self.device)),
                    f"ptts_t5_load_weight({name})")
            torch.cuda.current_stream(self.device).synchronize()  # the engine re-packs asynchronously; `t` must stay alive until then
        N.check(self.lib.ptts_t5_weights_ready(self._h), "ptts_t5_weights_ready")

    def graph_nodes(self) -> int:
        """Kernel nodes of the encoder graph captured last (``ptts_t5_debug_graph_nodes``; 0 before the first capture)."""
        n = C.c_int32(0)
        N.check(self.lib.ptts_t5_debug_graph_nodes(self._h, C.byref(n)), "ptts_t5_debug_graph_nodes")
        return int(n.value)

    def encode(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """input_ids int64 [B, N], attention_mask [B, N] or None -> float32 [B, N, d_model]: ``last_hidden_state`` with the masked
        positions zeroed (what generate() hands to the decoder, :3093-3097). Asynchronous on the current stream of the engine's device."""
        if input_ids.dim() != 2:
            raise ValueError(f"input_ids must be [batch, tokens], got {tuple(input_ids.shape)}")
        ids = input_ids.to(self.device, torch.int64).contiguous()
        B, Nn = ids.shape
        mk = attention_mask.to(self.device, torch.int32).contiguous() if attention_mask is not None else None
        if mk is not None and mk.shape != ids.shape:
            raise ValueError(f"attention_mask {tuple(mk.shape)} does not match input_ids {tuple(ids.shape)}")
        out = torch.empty(B, Nn, self.d_model, dtype=torch.float32, device=self.device)
        N.check(self.lib.ptts_t5_encode(self._h, C.c_void_p(ids.data_ptr()), C.c_void_p(mk.data_ptr()) if mk is not None else C.c_void_p(), B, Nn,
                                        C.c_void_p(out.data_ptr()), _stream_ptr(device=self.device)), "ptts_t5_encode")
        self._keep = (ids, mk)  # consumed asynchronously by the enqueued copies
        return out


class DacEngine:
    """Owner of a ``ptts_dac``: DAC codes → waveform (and, with ``encoder_dim`` > 0, waveform → codes for voice prompts)
    on MI355X (exact-f32 MFMA implicit-GEMM convolutions)."""

    def __init__(self, *, num_codebooks: int = 9, codebook_size: int = 1024, codebook_dim: int = 8, latent_dim: int = 1024,
                 decoder_dim: int = 1536, rates: Iterable[int] = (8, 8, 4, 2), max_batch: int = 1, max_frames: int = 2600,
                 device: Optional[torch.device] = None, encoder_dim: int = 0, compute_dtype: torch.dtype = torch.float32):
        if not torch.cuda.is_available():
            raise N.NativeLibraryError("DacEngine needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = N.load_library()
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        rates = tuple(int(r) for r in rates)
        arr = (C.c_int32 * 8)(*(list(rates) + [0] * (8 - len(rates))))
        if compute_dtype not in (torch.float32, torch.bfloat16):
            raise NotImplementedError(f"DAC compute dtype {compute_dtype} is not supported (float32: exact-f32 MFMA, bfloat16: bf16 MFMA operands)")
        self.compute_dtype = compute_dtype
        self.cfg = N.PttsDacConfig(num_codebooks, codebook_size, codebook_dim, latent_dim, decoder_dim, len(rates), arr,
                                   N.PTTS_BF16 if compute_dtype == torch.bfloat16 else N.PTTS_F32,
                                   max_batch, max_frames, self.device.index or 0, int(encoder_dim))
        self.encoder_dim = int(encoder_dim)
        self.latent_dim = latent_dim
        self.hop = math.prod(rates)
        self.K = num_codebooks
        self.codebook_size = codebook_size
        self.max_batch, self.max_frames = max_batch, max_frames
        self._h = C.c_void_p()
        N.check(self.lib.ptts_dac_create(C.byref(self.cfg), C.byref(self._h)), "ptts_dac_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.ptts_dac_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = ""):
        """`sd`: dac.model.DAC names with weight-norm in any of the three formats the reference may hold
        (folded ``.weight``, legacy ``.weight_g/.weight_v``, ``.parametrizations.weight.original0/1``;
        dac_wrapper/modeling_dac.py:148-164). Folded here on the host side once."""
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        for name, t in fold_weight_norm(sd).items():
            if not (name.startswith("quantizer.") or name.startswith("decoder.") or (self.encoder_dim > 0 and name.startswith("encoder."))):
                continue
            t = t.detach().to(self.device, torch.float32).contiguous()
            if name.endswith(".alpha"):
                t = t.reshape(-1).contiguous()
            N.check(self.lib.ptts_dac_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), _shape_arr(t), t.dim(), _stream_ptr(device=self.device)),
                    f"ptts_dac_load_weight({name})")
            torch.cuda.current_stream(self.device).synchronize()
        N.check(self.lib.ptts_dac_weights_ready(self._h), "ptts_dac_weights_ready")

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes int64 [B, K, T] → waveform float32 [B, 1, hop*T]."""
        if codes.dim() != 3 or codes.shape[1] != self.K:
            raise ValueError(f"audio_codes must be [batch, {self.K}, frames], got {tuple(codes.shape)}")
        codes = codes.to(self.device, torch.int64).contiguous()
        B, _, T = codes.shape
        out = torch.empty(B, 1, self.hop * T, dtype=torch.float32, device=self.device)
        N.check(self.lib.ptts_dac_decode(self._h, C.c_void_p(codes.data_ptr()), C.c_void_p(out.data_ptr()), B, T, _stream_ptr(device=self.device)), "ptts_dac_decode")
        self._keep = codes
        return out


    def compact_codes(self, codes: torch.Tensor):
        """codes int64 [B, K, T] → (codes with every frame holding an id outside [0, codebook_size) dropped, the kept ones in order
        [B, K, T]; kept-frame counts int32 [B]), both on the device (``ptts_dac_compact_codes``; modeling_parler_tts.py:3627-3636)."""
        if codes.dim() != 3 or codes.shape[1] != self.K:
            raise ValueError(f"audio_codes must be [batch, {self.K}, frames], got {tuple(codes.shape)}")
        codes = codes.to(self.device, torch.int64).contiguous()
        B, _, T = codes.shape
        out = torch.empty_like(codes)
        frames = torch.empty(B, dtype=torch.int32, device=self.device)
        N.check(self.lib.ptts_dac_compact_codes(self._h, C.c_void_p(codes.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(frames.data_ptr()), B, T,
                                                _stream_ptr(device=self.device)), "ptts_dac_compact_codes")
        self._keep = codes
        return out, frames

    def decode_ragged(self, codes: torch.Tensor, frames: torch.Tensor) -> torch.Tensor:
        """codes int64 [B, K, T], frames int32 [B] on the device → waveform float32 [B, 1, hop*T] in which utterance b is the decode of its
        first frames[b] frames and zero beyond (``ptts_dac_decode_ragged``: one pass, no per-utterance launches)."""
        if codes.dim() != 3 or codes.shape[1] != self.K:
            raise ValueError(f"audio_codes must be [batch, {self.K}, frames], got {tuple(codes.shape)}")
        codes = codes.to(self.device, torch.int64).contiguous()
        frames = frames.to(self.device, torch.int32).contiguous()
        B, _, T = codes.shape
        if frames.shape != (B,):
            raise ValueError(f"frames must be [batch], got {tuple(frames.shape)}")
        out = torch.empty(B, 1, self.hop * T, dtype=torch.float32, device=self.device)
        N.check(self.lib.ptts_dac_decode_ragged(self._h, C.c_void_p(codes.data_ptr()), C.c_void_p(frames.data_ptr()), C.c_void_p(out.data_ptr()), B, T,
                                                _stream_ptr(device=self.device)), "ptts_dac_decode_ragged")
        self._keep = (codes, frames)
        return out

    def decode_chunk(self, codes: torch.Tensor, first_frame: int, n_frames: int, halo: int, out: Optional[torch.Tensor] = None,
                     n_emit: Optional[int] = None) -> torch.Tensor:
        """codes int64 [B, K, T] (contiguous, on the device; read in place): decodes the window [first_frame - halo,
        first_frame + n_frames) and returns the samples of frames [first_frame, first_frame + n_emit) (default: all n_frames) as
        float32 [B, 1, hop*n_emit] (``ptts_dac_decode_chunk``). ``out``: a float32 [B, >= hop*(first_frame + n_emit)] waveform
        buffer to write into at sample offset hop*first_frame instead (chunked decode of a whole utterance). Enqueued on the
        CURRENT stream of the engine's device: a caller can run it on a side stream while the decoder graph replays."""
        if codes.dim() != 3 or codes.shape[1] != self.K:
            raise ValueError(f"audio_codes must be [batch, {self.K}, frames], got {tuple(codes.shape)}")
        if codes.device != self.device or codes.dtype != torch.int64 or not codes.is_contiguous():
            codes = codes.to(self.device, torch.int64).contiguous()
        B, _, T = codes.shape
        n_emit = n_frames if n_emit is None else int(n_emit)
        if out is None:
            dst = torch.empty(B, 1, self.hop * n_emit, dtype=torch.float32, device=self.device)
            ptr, ld = dst.data_ptr(), self.hop * n_emit
        else:
            if out.dtype != torch.float32 or out.dim() != 2 or out.shape[0] != B or out.stride(1) != 1 or out.shape[1] < self.hop * (first_frame + n_emit):
                raise ValueError("out must be a float32 [batch, samples] buffer covering the emitted frames")
            dst, ptr, ld = out, out.data_ptr() + 4 * self.hop * first_frame, out.stride(0)
        N.check(self.lib.ptts_dac_decode_chunk(self._h, C.c_void_p(codes.data_ptr()), T, int(first_frame), int(n_frames), int(halo),
                                               C.c_void_p(ptr), int(ld), int(n_emit), B, _stream_ptr(device=self.device)), "ptts_dac_decode_chunk")
        self._keep = codes
        return dst

    def encode(self, wave: torch.Tensor, n_quantizers: Optional[int] = None) -> torch.Tensor:
        """waveform float32 [B, 1, L] (L a multiple of the hop) → codes int64 [B, n_quantizers, L / hop]."""
        if self.encoder_dim <= 0:
            raise NotImplementedError("this DacEngine was built without the encoder (encoder_dim=0)")
        if wave.dim() != 3 or wave.shape[1] != 1:
            raise ValueError(f"input_values must be [batch, 1, samples], got {tuple(wave.shape)}")
        wave = wave.to(self.device, torch.float32).contiguous()
        B, _, L = wave.shape
        if L == 0 or L % self.hop:
            raise ValueError(f"waveform length {L} is not a positive multiple of the hop {self.hop} (apply the preprocess padding)")
        nq = self.K if n_quantizers is None else max(1, min(int(n_quantizers), self.K))
        codes = torch.empty(B, nq, L // self.hop, dtype=torch.int64, device=self.device)
        N.check(self.lib.ptts_dac_encode(self._h, C.c_void_p(wave.data_ptr()), C.c_void_p(codes.data_ptr()), B, L, nq, _stream_ptr(device=self.device)), "ptts_dac_encode")
        self._keep = wave
        return codes

    def debug_latents(self, B: int, T: int) -> torch.Tensor:
        """Latents z of the last ``encode`` as [B, latent, T] (parity probe)."""
        p = C.c_void_p()
        N.check(self.lib.ptts_dac_debug_latents(self._h, C.byref(p)), "ptts_dac_debug_latents")
        out = torch.empty(B, T, self.latent_dim, dtype=torch.float32, device=self.device)
        rc = N.hip_runtime().hipMemcpyAsync(C.c_void_p(out.data_ptr()), C.c_void_p(p.value), C.c_size_t(out.numel() * 4), 3, _stream_ptr(device=self.device))
        if rc != 0:
            raise N.NativeLibraryError(f"hipMemcpy(D2D) failed with code {rc}")
        return out.transpose(1, 2).contiguous()
    def debug_stage(self, codes: torch.Tensor, stage: int):
        """Parity probe (``ptts_dac_debug_decode_upto``): decode stopped after `stage`; returns (act [B, C, rows] float32 - the exact values
        of the bf16 / fp32 activation buffer -, raw [B, C, rows] float32 residual stream or None)."""
        codes = codes.to(self.device, torch.int64).contiguous()
        B, _, T = codes.shape
        act, raw = C.c_void_p(), C.c_void_p()
        is_bf, rows, ch = C.c_int32(), C.c_int32(), C.c_int32()
        N.check(self.lib.ptts_dac_debug_decode_upto(self._h, C.c_void_p(codes.data_ptr()), B, T, int(stage), _stream_ptr(device=self.device), C.byref(act),
                                                    C.byref(is_bf), C.byref(raw), C.byref(rows), C.byref(ch)), "ptts_dac_debug_decode_upto")
        hip = N.hip_runtime()

        def grab(ptr, dtype):
            t = torch.empty(B, rows.value, ch.value, dtype=dtype, device=self.device)
            rc = hip.hipMemcpyAsync(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), C.c_size_t(t.numel() * t.element_size()), 3, _stream_ptr(device=self.device))
            if rc != 0:
                raise N.NativeLibraryError(f"hipMemcpy(D2D) failed with code {rc}")
            return t.float().transpose(1, 2).contiguous()

        a = grab(act.value, torch.bfloat16 if is_bf.value else torch.float32)
        r = grab(raw.value, torch.float32) if raw.value else None
        return a, r


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """w = g · v / ‖v‖ (norm over all dims but 0, torch weight_norm's default dim=0) for both key formats."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        for g_suf, v_suf in ((".weight_g", ".weight_v"), (".parametrizations.weight.original0", ".parametrizations.weight.original1")):
            if k.endswith(g_suf):
                base = k[: -len(g_suf)]
                vv = sd[base + v_suf].float()
                norm = vv.flatten(1).norm(dim=1).view(-1, *([1] * (vv.dim() - 1)))
                out[base + ".weight"] = v.float() * vv / norm
                break
            if k.endswith(v_suf):
                break
        else:
            out[k] = v
    return out
