"""Host side of the MI355X-native generation path: the reference's public surface
(``ParlerTTSForConditionalGeneration.from_pretrained()/.generate()``, parler_tts/modeling_parler_tts.py:2306-3678)
re-implemented around the two HIP engines. Nothing here inherits from or imports the reference; the `_sample`
loop of transformers (which the reference delegates to, :3564) is owned here and, on the default path, runs
entirely on the device (engine.generate_ids): Python only encodes the description (stock PyTorch-ROCm T5, third
party and off the per-token loop), hands pointers to the engine, un-delays the ids and calls the DAC engine.

State-dict names are the reference's: ``text_encoder.*``, ``audio_encoder.model.*``,
``decoder.model.decoder.layers.N.*``, ``decoder.lm_heads.N.*``, ``embed_prompts.*``, ``enc_to_dec_proj.*``.
"""
from __future__ import annotations

import copy
import json
import math
import os
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from .configuration_parler_tts import DACConfig, ParlerTTSConfig, ParlerTTSDecoderConfig
from .dac_wrapper import DACModel
from .engine import DecoderEngine, T5Engine
from .generation_extras import active_extras, build_processors, build_stopping_criteria, check_generation_mode, check_model_kwargs
from .logits_processors import ParlerTTSLogitsProcessor


# ------------------------------------------------------------------------------------------------------------
# delay pattern (reference :205-276) — closed form instead of the reference's loop + triu/tril
# ------------------------------------------------------------------------------------------------------------
def build_delay_pattern_mask(input_ids: torch.LongTensor, bos_token_id: int, pad_token_id: int, max_length: int,
                             num_codebooks: int) -> Tuple[torch.LongTensor, torch.LongTensor]:
    """Codebook k is delayed by k steps: position j of row k holds BOS for j <= k, PAD for j - k >= max_length - K + 1,
    the (shifted) prompt where one was given, and -1 ("to be predicted") elsewhere. Returns (ids up to the first
    position that needs predicting, full pattern mask)."""
    dev = input_ids.device
    ids = input_ids.reshape(-1, num_codebooks, input_ids.shape[-1])
    bsz, K, seq_len = ids.shape
    pattern = torch.full((bsz, K, max_length), -1, dtype=torch.long, device=dev)
    if max_length < 2 * K - 1:  # too short for the pattern: returned as is (:246-247)
        return ids.reshape(bsz * K, -1), pattern.reshape(bsz * K, -1)
    j = torch.arange(max_length, device=dev)[None, :]
    k = torch.arange(K, device=dev)[:, None]
    src = j - k  # column of the un-shifted prompt that lands on (k, j)
    has_prompt = (src >= 0) & (src < seq_len)
    gathered = torch.gather(ids, 2, src.clamp(0, seq_len - 1)[None].expand(bsz, -1, -1))
    pattern = torch.where(has_prompt[None], gathered, pattern)
    bos = torch.as_tensor(bos_token_id, device=dev, dtype=torch.long)
    pad = torch.as_tensor(pad_token_id, device=dev, dtype=torch.long)
    pattern = torch.where((j <= k)[None], bos, pattern)
    pattern = torch.where(((j - k) >= max_length - K + 1)[None], pad, pattern)
    first_row = pattern[:, 0, :]
    todo = (first_row == -1).nonzero()
    first_start = int(todo[:, 1].min()) if todo.numel() > 0 else seq_len
    mask = pattern.reshape(bsz * K, -1)
    return pattern[..., :first_start].reshape(bsz * K, -1), mask


def apply_delay_pattern_mask(input_ids: torch.LongTensor, decoder_pad_token_mask: torch.LongTensor) -> torch.LongTensor:
    m = decoder_pad_token_mask[..., : input_ids.shape[-1]]
    return torch.where(m == -1, input_ids, m)


# ------------------------------------------------------------------------------------------------------------
# weight holders that reproduce the reference's parameter names
# ------------------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    pass


def _set_param(root: nn.Module, dotted: str, tensor: torch.Tensor):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Holder())
        m = m._modules[p]
    m.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


def sinusoidal_table(num_embeddings: int, embedding_dim: int) -> torch.Tensor:
    """cos ‖ sin halves with frequencies exp(-i·ln(1e4)/(half-1)) (reference :346-359)."""
    half = embedding_dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.int64).float() * -(math.log(10000) / (half - 1)))
    ang = torch.arange(num_embeddings, dtype=torch.int64).float().unsqueeze(1) * freq.unsqueeze(0)
    tab = torch.cat([torch.cos(ang), torch.sin(ang)], dim=1).view(num_embeddings, -1)
    if embedding_dim % 2 == 1:
        tab = torch.cat([tab, torch.zeros(num_embeddings, 1)], dim=1)
    return tab


class ParlerTTSForCausalLM(nn.Module):
    """Holder of the decoder-LM parameters under the reference's names (``model.decoder.*``, ``lm_heads.*``).
    The arithmetic of ``ParlerTTSForCausalLM.forward`` (reference :1865) lives in the HIP engine."""

    config_class = ParlerTTSDecoderConfig

    def __init__(self, config: ParlerTTSDecoderConfig, init_weights: bool = True):
        super().__init__()
        self.config = config
        self.num_codebooks = config.num_codebooks
        c, H, F = config, config.hidden_size, config.ffn_dim
        std = c.initializer_factor
        mk = (lambda *s: torch.randn(*s) * std) if init_weights else (lambda *s: torch.empty(*s))
        for k in range(c.num_codebooks):
            _set_param(self, f"model.decoder.embed_tokens.{k}.weight", mk(c.vocab_size + 1, H))
        if not c.rope_embeddings:
            _set_param(self, "model.decoder.embed_positions.weights", sinusoidal_table(c.max_position_embeddings, H))
        for i in range(c.num_hidden_layers):
            lp = f"model.decoder.layers.{i}."
            hd = H // c.num_attention_heads
            for att, nkv in (("self_attn", c.num_key_value_heads), ("encoder_attn", c.num_cross_attention_key_value_heads)):
                for proj in ("k_proj", "v_proj", "q_proj", "out_proj"):  # grouped-query attention: K/V projections have nkv * head_dim rows (:449-452)
                    _set_param(self, f"{lp}{att}.{proj}.weight", mk(nkv * hd if proj in ("k_proj", "v_proj") else H, H))
                _set_param(self, f"{lp}{att}_layer_norm.weight", torch.ones(H))
                _set_param(self, f"{lp}{att}_layer_norm.bias", torch.zeros(H))
            _set_param(self, f"{lp}fc1.weight", mk(F, H))
            _set_param(self, f"{lp}fc2.weight", mk(H, F))
            _set_param(self, f"{lp}final_layer_norm.weight", torch.ones(H))
            _set_param(self, f"{lp}final_layer_norm.bias", torch.zeros(H))
        _set_param(self, "model.decoder.layer_norm.weight", torch.ones(H))
        _set_param(self, "model.decoder.layer_norm.bias", torch.zeros(H))
        if c.use_fused_lm_heads:
            _set_param(self, "lm_heads.weight", mk(c.num_codebooks * c.vocab_size, H))
        else:
            for k in range(c.num_codebooks):
                _set_param(self, f"lm_heads.{k}.weight", mk(c.vocab_size, H))

    def build_delay_pattern_mask(self, input_ids, bos_token_id, pad_token_id, max_length=None):
        if max_length is None:  # :2068: the model's generation_config (a bare PreTrainedModel default: max_length 20)
            gc = getattr(self, "generation_config", None)
            max_length = int(gc.max_length) if gc is not None and getattr(gc, "max_length", None) is not None else 20
        return build_delay_pattern_mask(input_ids, bos_token_id, pad_token_id, max_length, self.num_codebooks)

    def save_pretrained(self, save_directory: str, safe_serialization: bool = True, **kwargs):
        """config.json + model.safetensors with the reference's decoder key names (helpers/model_init_scripts/init_model_600M.py:46-47)."""
        from safetensors.torch import save_file

        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}, os.path.join(save_directory, "model.safetensors"),
                  metadata={"format": "pt"})

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, *model_args, config=None, **kwargs):
        """A decoder directory written by ``save_pretrained`` - or a full Parler-TTS checkpoint, whose ``decoder.*`` tensors are taken
        (:2654-2666 accepts both)."""
        from safetensors.torch import load_file

        path = pretrained_model_name_or_path
        if config is None:
            with open(os.path.join(path, "config.json")) as fh:
                d = json.load(fh)
            config = ParlerTTSConfig.from_dict(d).decoder if "decoder" in d else ParlerTTSDecoderConfig.from_dict(d)
        if isinstance(config, ParlerTTSConfig):
            config = config.decoder
        m = cls(config, init_weights=False)
        sd = load_file(os.path.join(path, "model.safetensors"))
        if any(k.startswith("decoder.") for k in sd):
            sd = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
        missing, unexpected = m.load_state_dict(sd, strict=False)
        missing = [k for k in missing if "rotary_emb" not in k]
        if missing:
            raise RuntimeError(f"decoder checkpoint {path} lacks {missing[:5]}")
        return m

    @staticmethod
    def apply_delay_pattern_mask(input_ids, decoder_pad_token_mask):
        return apply_delay_pattern_mask(input_ids, decoder_pad_token_mask)

    # accessors of the reference class (:1845-1861); the modules they return are parameter holders with the reference's names
    def get_input_embeddings(self):
        return self.model.decoder.embed_tokens

    def set_input_embeddings(self, value):
        self.model.decoder.embed_tokens = value

    def get_output_embeddings(self):
        return self.lm_heads

    def set_output_embeddings(self, new_embeddings):
        self.lm_heads = new_embeddings

    def get_decoder(self):
        return self.model.decoder

    def set_decoder(self, decoder):
        self.model.decoder = decoder

    def forward(self, *args, **kwargs):
        raise NotImplementedError("the decoder forward runs inside the HIP engine; call ParlerTTSForConditionalGeneration.generate()")

    def generate(self, *args, **kwargs):
        """The reference's decoder-only ``generate`` (:2072-2300, inherited from MusicGen) cannot run there either: it calls
        ``build_delay_pattern_mask(input_ids, pad_token_id=..., max_length=...)`` without the required ``bos_token_id`` (:2189-2193 vs the
        signature at :2041-2043) and raises TypeError. Generation goes through the composite model."""
        raise NotImplementedError("decoder-only generate() is dead code in the reference (TypeError at modeling_parler_tts.py:2189); "
                                  "call ParlerTTSForConditionalGeneration.generate()")


class GenerateOutput(dict):
    """``return_dict_in_generate=True``: ``.sequences`` is the waveform, ``["audios_length"]`` the lengths (:3648-3651)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _single_device(device_map):
    """``device_map`` forms that name ONE device (the whole model lives on one GPU: SURVEY.md §8(e), one process per GPU); a real
    multi-device map has no meaning for this engine and is refused."""
    if device_map is None:
        return None
    if isinstance(device_map, dict):
        targets = set(device_map.values())
        if len(targets) != 1:
            raise NotImplementedError(f"device_map={device_map}: the HIP engines keep a whole model replica on one device (one process per GPU)")
        device_map = targets.pop()
    if device_map in ("auto", "balanced", "sequential"):
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if isinstance(device_map, int):
        return torch.device("cuda", device_map)
    return torch.device(device_map)


def _build_uninitialised(factory):
    """Runs a module factory with transformers' weight initialisation switched off (the tensors are allocated, their values are
    whatever the allocator returned): for replicas whose weights arrive right afterwards. Falls back to the plain factory on
    transformers releases without the context manager."""
    try:
        from transformers.initialization import no_init_weights
    except Exception:  # noqa: BLE001
        try:
            from transformers.modeling_utils import no_init_weights
        except Exception:  # noqa: BLE001
            return factory()
    with no_init_weights():
        module = factory()
    # weight tying is part of the skipped initialisation in some transformers releases (T5: encoder.embed_tokens <-> shared): redo it,
    # so that every replica has the SAME parameter list as an initialised one (the weight broadcast walks model.parameters())
    if hasattr(module, "tie_weights"):
        try:
            module.tie_weights()
        except Exception:  # noqa: BLE001 - a release whose tie needs the initialised path: build it the ordinary (initialised) way
            return factory()  # never hand out a module whose tied embedding may be unallocated garbage under one of its names
    return module


def _default_generation_config(config: ParlerTTSConfig):
    from transformers import GenerationConfig

    d = config.decoder
    gc = GenerationConfig(max_length=int(30 * config.audio_encoder.frame_rate), do_sample=True, pad_token_id=d.pad_token_id,
                          eos_token_id=d.eos_token_id, bos_token_id=d.bos_token_id, decoder_start_token_id=d.bos_token_id)
    return gc  # helpers/model_init_scripts/init_model_600M.py:57-63


class _StepPump:
    """Keeps the decoder graph enqueued at most `max_ahead` steps beyond the oldest chunk boundary still in flight, in pieces of
    a few steps, polling that boundary's event between pieces. The look-ahead keeps the GPU busy while the host forwards the
    finished chunk; it is bounded because (a) hipGraphLaunch blocks the calling thread once the hardware queue is full and
    (b) HIP multiplexes streams onto a few hardware queues: when the side stream shares one with the main stream, its copies
    and codec kernels run only after everything already enqueued there, so the first audio is late by the look-ahead."""

    def __init__(self, eng, stream, first: int, chunk: int, remaining: int, piece: int = 4, max_ahead: int = 16):
        self.eng, self.stream, self.chunk, self.piece, self.max_ahead = eng, stream, chunk, piece, max_ahead
        self.remaining = remaining
        self.cur_left = first        # steps still to enqueue before the next boundary; -1 = nothing left to enqueue
        self.in_flight = []          # events of the boundaries enqueued and not yet observed (oldest first)
        self.after = []              # steps enqueued after in_flight[i] (and before in_flight[i + 1])
        self.enqueued = 0            # decode steps enqueued so far
        self.at_boundary = []        # value of `enqueued` when in_flight[i] was recorded
        self.done_steps = 0          # decode steps covered by the last boundary wait_boundary() observed complete

    def _enqueue_piece(self):
        n = min(self.piece, self.cur_left)
        if n > 0:
            self.eng.decode_steps(n)
            self.cur_left -= n
            self.remaining -= n
            self.enqueued += n
            if self.after:
                self.after[-1] += n
        if self.cur_left == 0:
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self.in_flight.append(ev)
            self.after.append(0)
            self.at_boundary.append(self.enqueued)
            self.cur_left = min(self.chunk, self.remaining) if self.remaining > 0 else -1

    def stop(self):
        """every row finished: steps already enqueued are device-side no-ops, enqueue no more"""
        self.cur_left, self.remaining = -1, 0

    def wait_boundary(self) -> bool:
        """True once the oldest boundary in flight has completed; False when nothing is in flight and nothing is left."""
        while True:
            if self.in_flight and self.in_flight[0].query():
                break
            if self.cur_left >= 0 and sum(self.after) < self.max_ahead:
                self._enqueue_piece()
                continue
            if not self.in_flight:
                return False
            self.in_flight[0].synchronize()
            break
        self.in_flight.pop(0)
        self.after.pop(0)
        self.done_steps = self.at_boundary.pop(0)
        self.max_ahead = max(self.max_ahead, self.chunk)  # only the FIRST boundary is latency-critical: a full chunk ahead afterwards
        return True


class ParlerTTSForConditionalGeneration(nn.Module):
    config_class = ParlerTTSConfig
    base_model_prefix = "encoder_decoder"
    main_input_name = "input_ids"

    def __init__(self, config: Optional[ParlerTTSConfig] = None, text_encoder: Optional[nn.Module] = None,
                 audio_encoder: Optional[nn.Module] = None, decoder: Optional[ParlerTTSForCausalLM] = None, init_weights: bool = True):
        """``init_weights=False``: allocate the parameters without drawing them (a replica that is about to receive its weights from
        a checkpoint or from rank 0's broadcast: the random init of the 1.2 G parameters of Mini-v1 + T5 takes ~15-25 s of host time)."""
        super().__init__()
        if config is None and (text_encoder is None or audio_encoder is None or decoder is None):
            raise ValueError("Either a configuration has to be provided, or all three of text encoder, audio encoder and Parler-TTS decoder.")
        if config is None:
            config = ParlerTTSConfig.from_sub_models_config(text_encoder.config, audio_encoder.config, decoder.config)
        elif not isinstance(config, self.config_class):
            raise ValueError(f"Config: {config} has to be of type {self.config_class}")
        xh = getattr(config.decoder, "cross_attention_hidden_size", None)
        if xh is not None and xh != config.text_encoder.hidden_size:
            raise ValueError("If `cross_attention_hidden_size` is specified in the Parler-TTS decoder's configuration, it has to be equal"
                             f" to the text encoder's `hidden_size`. Got {xh} and {config.text_encoder.hidden_size}.")
        config.decoder.check_supported_by_engine()
        self.config = config
        if text_encoder is None:
            from transformers import AutoModelForTextEncoding

            if init_weights:
                text_encoder = AutoModelForTextEncoding.from_config(config.text_encoder)  # third-party T5 encoder (:2345-2348)
            else:
                text_encoder = _build_uninitialised(lambda: AutoModelForTextEncoding.from_config(config.text_encoder))
        self.text_encoder = text_encoder
        self.audio_encoder = audio_encoder if audio_encoder is not None else DACModel(config.audio_encoder)
        self.decoder = decoder if decoder is not None else ParlerTTSForCausalLM(config.decoder, init_weights=init_weights)
        H = config.decoder.hidden_size
        if config.text_encoder.hidden_size != H and xh is None:
            self.enc_to_dec_proj = nn.Linear(config.text_encoder.hidden_size, H)  # :2388-2392
        self.embed_prompts = nn.Embedding(config.vocab_size, H)  # :2395
        if init_weights:
            self.embed_prompts.weight.data.normal_(mean=0.0, std=config.decoder.initializer_factor)
        self.prompt_cross_attention = config.prompt_cross_attention
        if config.prompt_cross_attention:
            self.embed_positions = _Holder()
            _set_param(self.embed_positions, "weights", sinusoidal_table(config.decoder.max_position_embeddings, H))
        self.use_audio_scales = True      # DACModel.decode takes `audio_scales` (:2416-2417)
        self.use_4dim_audio_codes = True  # model_type "dac_on_the_hub" (:2419-2422)
        self.generation_config = _default_generation_config(config)
        self._engine = None  # (property) drops every cached decoder engine
        # the HIP engines hold PACKED COPIES of the decoder's and the text encoder's weights: loading a state dict into one of those sub-modules
        # directly (model.text_encoder.load_state_dict(...), not through this class's load_state_dict) must drop the copies too
        import weakref

        me = weakref.ref(self)

        def _drop_engines(module, incompatible_keys):
            m = me()
            if m is not None:
                m._engine = None

        for sub in (self.decoder, self.text_encoder):
            if hasattr(sub, "register_load_state_dict_post_hook"):
                sub.register_load_state_dict_post_hook(_drop_engines)
        for p in self.parameters():
            p.requires_grad_(False)
        self.eval()

    # -- bookkeeping ---------------------------------------------------------------------------------------
    @property
    def device(self) -> torch.device:
        return self.embed_prompts.weight.device

    @property
    def dtype(self) -> torch.dtype:
        return self.embed_prompts.weight.dtype

    def get_text_encoder(self):
        return self.text_encoder

    def get_audio_encoder(self):
        return self.audio_encoder

    def get_encoder(self):
        return self.get_text_encoder()

    def get_decoder(self):
        return self.decoder

    def get_input_embeddings(self):
        return self.text_encoder.get_input_embeddings()

    def get_output_embeddings(self):
        return self.decoder.lm_heads

    def set_output_embeddings(self, new_embeddings):
        self.decoder.lm_heads = new_embeddings
        self._engine = None

    def forward(self, *args, **kwargs):
        raise NotImplementedError("the training forward (:2695-2880) is outside this inference implementation; call .generate()")

    def prepare_decoder_input_ids_from_labels(self, labels: torch.Tensor):
        """shift_tokens_right over [bsz, seq, codebooks] labels, transposed to [bsz, codebooks, seq] (:3196-3199, :187-202)."""
        d = self.config.decoder
        labels = labels.transpose(1, 2)
        shifted = labels.new_zeros(labels.shape)
        shifted[..., 1:] = labels[..., :-1].clone()
        shifted[..., 0] = d.bos_token_id
        return shifted.masked_fill(shifted == -100, d.pad_token_id)

    def freeze_encoders(self, freeze_text_encoder=True):
        pass  # inference-only implementation: nothing is trainable

    def resize_token_embeddings(self, *args, **kwargs):
        raise NotImplementedError("Resizing the embedding layers via the EncoderDecoderModel directly is not supported. Please use the"
                                  " respective methods of the wrapped objects (model.encoder.resize_token_embeddings(...) or"
                                  " model.decoder.resize_token_embeddings(...))")

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._engine = None  # device / dtype may have changed: repack lazily
        self.__dict__.pop("_enc_graphs", None)  # captured encoder graphs point at the old parameter storage
        return out

    # -- (de)serialisation ----------------------------------------------------------------------------------
    def save_pretrained(self, save_directory: str, safe_serialization: bool = True, **kwargs):
        from safetensors.torch import save_file

        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        with open(os.path.join(save_directory, "generation_config.json"), "w") as f:
            json.dump({k: v for k, v in self.generation_config.to_dict().items() if v is not None}, f, indent=2, default=str)
        sd, seen = {}, set()
        for k, v in self.state_dict().items():  # tied tensors (T5 shared embedding) are written once, like HF does
            if k.endswith("_dummy") or v.data_ptr() in seen:
                continue
            seen.add(v.data_ptr())
            sd[k] = v.detach().cpu().contiguous()
        sd.update({"audio_encoder." + k: v.detach().cpu().contiguous() for k, v in self.audio_encoder.state_dict().items()})
        save_file(sd, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, *model_args, torch_dtype: Optional[torch.dtype] = None,
                        attn_implementation: Optional[str] = None, config: Optional[ParlerTTSConfig] = None, **kwargs):
        """Loads a HF checkpoint directory (config.json + [generation_config.json] + *.safetensors) with the
        reference's key names. ``attn_implementation`` is accepted for drop-in compatibility and ignored: the
        whole attention registry (:933-937) is replaced by the HIP attention kernel."""
        path = pretrained_model_name_or_path
        if torch_dtype is None and kwargs.get("dtype") is not None:
            torch_dtype = kwargs["dtype"]  # newer transformers releases spell it `dtype=`
        if isinstance(torch_dtype, str):
            torch_dtype = None if torch_dtype == "auto" else getattr(torch, torch_dtype)
        if torch_dtype is not None and torch_dtype not in (torch.float32, torch.bfloat16):
            raise NotImplementedError(f"torch_dtype={torch_dtype}: the HIP engine implements float32 (parity) and bfloat16 (throughput)")
        device = _single_device(kwargs.get("device_map"))
        # hub / loader plumbing that has no effect on a locally materialised state dict, everything else must be a config field or raises
        loader_kw = {"dtype", "device_map", "revision", "cache_dir", "token", "local_files_only", "force_download", "resume_download", "proxies",
                     "subfolder", "variant", "use_safetensors", "low_cpu_mem_usage", "trust_remote_code", "use_auth_token", "weights_only",
                     "output_loading_info", "offload_folder", "offload_state_dict", "max_memory", "tp_plan"}
        overrides = {k: v for k, v in kwargs.items() if k not in loader_kw}
        if not os.path.isdir(path):
            from huggingface_hub import snapshot_download  # needs network / a local HF cache

            hub_kw = {k: kwargs[k] for k in ("revision", "cache_dir", "token", "local_files_only") if kwargs.get(k) is not None}
            path = snapshot_download(path, allow_patterns=["*.json", "*.safetensors"], **hub_kw)
            if not any(f.endswith(".safetensors") for f in os.listdir(path)):  # legacy repos ship pytorch_model.bin only
                path = snapshot_download(pretrained_model_name_or_path, allow_patterns=["*.json", "*.bin"], **hub_kw)
        cfg = copy.deepcopy(config) if config is not None else ParlerTTSConfig.from_pretrained(path)  # overrides never touch the caller's object
        for k, v in overrides.items():  # transformers semantics: a kwarg naming a config attribute overrides it, any other is an error
            if not hasattr(cfg, k):
                raise TypeError(f"{cls.__name__}.from_pretrained() got an unexpected keyword argument '{k}'")
            setattr(cfg, k, v)
        model = cls(cfg, init_weights=False)  # every tensor comes from the checkpoint (strict load below): no ~15-25 s random init first
        gpath = os.path.join(path, "generation_config.json")
        if os.path.exists(gpath):
            from transformers import GenerationConfig

            model.generation_config = GenerationConfig.from_pretrained(path)
        from safetensors.torch import load_file

        index = os.path.join(path, "model.safetensors.index.json")
        sd: Dict[str, torch.Tensor] = {}
        if os.path.exists(index) or os.path.exists(os.path.join(path, "model.safetensors")):
            files = sorted(set(json.load(open(index))["weight_map"].values())) if os.path.exists(index) else ["model.safetensors"]
            for fn in files:
                sd.update(load_file(os.path.join(path, fn)))
        elif os.path.exists(os.path.join(path, "pytorch_model.bin")):  # legacy torch.save checkpoints (tensors only)
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        else:
            raise FileNotFoundError(f"no model.safetensors[.index.json] or pytorch_model.bin under {path}")
        model.load_state_dict(sd)
        if torch_dtype is not None:
            model.to(dtype=torch_dtype)
        if device is not None:
            model.to(device)
        return model

    @classmethod
    def from_sub_models_pretrained(cls, text_encoder_pretrained_model_name_or_path: Optional[str] = None,
                                   audio_encoder_pretrained_model_name_or_path: Optional[str] = None,
                                   decoder_pretrained_model_name_or_path: Optional[str] = None, *model_args, **kwargs):
        """Composes a model from three sub-model checkpoints (:2490-2691; used by helpers/model_init_scripts/*.py). Keyword
        arguments prefixed ``text_encoder_`` / ``audio_encoder_`` / ``decoder_`` go to the respective loader (``*_model`` passes
        an instantiated module, ``*_config`` a config); the rest updates the composite config (e.g. ``vocab_size``,
        ``prompt_cross_attention``)."""
        split = {"text_encoder": {}, "audio_encoder": {}, "decoder": {}}
        for k in list(kwargs):
            for pre in split:
                if k.startswith(pre + "_"):
                    split[pre][k[len(pre) + 1:]] = kwargs.pop(k)
                    break
        text_encoder = split["text_encoder"].pop("model", None)
        if text_encoder is None:
            if text_encoder_pretrained_model_name_or_path is None:
                raise ValueError("If `text_encoder_model` is not defined as an argument, a `text_encoder_pretrained_model_name_or_path` has to be defined.")
            from transformers import AutoModelForTextEncoding

            text_encoder = AutoModelForTextEncoding.from_pretrained(text_encoder_pretrained_model_name_or_path, *model_args, **split["text_encoder"])
        audio_encoder = split["audio_encoder"].pop("model", None)
        if audio_encoder is None:
            if audio_encoder_pretrained_model_name_or_path is None:
                raise ValueError("If `audio_encoder_model` is not defined as an argument, an `audio_encoder_pretrained_model_name_or_path` has to be defined.")
            audio_encoder = DACModel.from_pretrained(audio_encoder_pretrained_model_name_or_path, **split["audio_encoder"])
        decoder = split["decoder"].pop("model", None)
        if decoder is None:
            if decoder_pretrained_model_name_or_path is None:
                raise ValueError("If `decoder_model` is not defined as an argument, a `decoder_pretrained_model_name_or_path` has to be defined.")
            decoder = ParlerTTSForCausalLM.from_pretrained(decoder_pretrained_model_name_or_path, **split["decoder"])
        config = ParlerTTSConfig.from_sub_models_config(text_encoder.config, audio_encoder.config, decoder.config, **kwargs)
        return cls(text_encoder=text_encoder, audio_encoder=audio_encoder, decoder=decoder, config=config)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        audio = {k[len("audio_encoder."):]: v for k, v in state_dict.items() if k.startswith("audio_encoder.")}
        rest = {k: v for k, v in state_dict.items() if not k.startswith("audio_encoder.")}
        if audio:
            self.audio_encoder.load_state_dict(audio, strict=strict)
        own = super().state_dict()
        missing = [k for k in own if k not in rest and not k.startswith("audio_encoder.") and "rotary_emb" not in k]
        # T5 ties / registers a shared embedding under two names in some transformers releases: tolerate either
        missing = [k for k in missing if not (k.startswith("text_encoder.") and ("embed_tokens" in k or k.endswith("shared.weight")))]
        unexpected = [k for k in rest if k not in own and "rotary_emb" not in k]
        if strict and (missing or [k for k in unexpected if not k.startswith("text_encoder.")]):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        super().load_state_dict({k: v for k, v in rest.items() if k in own}, strict=False)
        # the T5 embedding exists under two names (shared / encoder.embed_tokens); a checkpoint may hold only one. Either the two are ONE
        # tensor (tied: loading one name filled both) or both names were in the checkpoint - otherwise one of them would keep whatever
        # the (possibly uninitialised, init_weights=False) allocation held
        te = getattr(self, "text_encoder", None)
        sh, emb = getattr(te, "shared", None), getattr(getattr(te, "encoder", None), "embed_tokens", None)
        if strict and sh is not None and emb is not None and sh.weight.data_ptr() != emb.weight.data_ptr():
            names = [k for k in rest if k.startswith("text_encoder.") and (k.endswith("shared.weight") or k.endswith("embed_tokens.weight"))]
            if 0 < len(names) < 2:
                raise RuntimeError(f"text_encoder embedding is not tied (shared / encoder.embed_tokens are different tensors) and the checkpoint "
                                   f"holds only {names}: the other copy would stay uninitialised")
        self._engine = None
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # -- engine ------------------------------------------------------------------------------------------------
    def enable_fp8_weights(self, enabled: bool = True):
        """Weight-only OCP e4m3 storage for the decoder's projection matrices (BASELINE configs[4]): the decode step at batch
        <= 4 streams 1-byte weights with one power-of-two scale per output row; the model must be in bfloat16. Not a reference
        feature (the reference has no quantised mode): outputs are those of the quantised model, checked against the oracle
        evaluating the SAME quantised weights (oracle/fp8_oracle.py)."""
        self.decoder_weights_fp8 = bool(enabled)
        self._engine = None
        return self

    def enable_fp8_kv_cache(self, enabled: bool = True):
        """Opt-in e4m3 self-attention KV cache for engines of more than 8 utterances (``ptts_config::kv_fp8``): 64 bytes + one power-of-two
        scale per (utterance, head, position) instead of 128 bytes, for 64+ utterances per GPU where the K/V stream bounds a third of the step.
        Not a reference feature (the reference raises on quantised caches, modeling_parler_tts.py:3497-3501): outputs are those of the model with
        a quantised cache, checked against the oracle applying the SAME quantiser (oracle/fp8_oracle.py: quantize_kv_rows). Smaller batches
        keep the bf16 cache."""
        self.decoder_kv_fp8 = bool(enabled)
        self._engine = None
        return self

    def _get_engine(self, B: int, N: int, P: int, max_length: int, T: int = 0) -> DecoderEngine:
        """T = voice-prompt frames: the prefill runs the P prompt positions, the BOS column and the T given columns in one pass."""
        dev, dt = self.device, self.dtype
        if dev.type != "cuda":
            raise RuntimeError("generate() runs on the HIP engine only: move the model to a cuda device first (there is no CPU fallback)")
        if dt not in (torch.float32, torch.bfloat16):
            raise NotImplementedError(f"model dtype {dt}: the HIP engine implements float32 (parity) and bfloat16 (throughput)")
        fp8 = bool(getattr(self, "decoder_weights_fp8", False))
        if fp8 and dt != torch.bfloat16:
            raise NotImplementedError("decoder_weights_fp8 needs the model in bfloat16 (e4m3 weights, bf16 activations)")
        # One engine per batch-size class, capacities grow-only inside a class. The engine tunes itself to its max_batch at creation: KV
        # splits of the self-attention (4 up to 4 utterances, 2 for 5..8, none above) and which weight copies it holds (row-major for the
        # GEMV step up to 8 utterances; MFMA strips only - bf16 or e4m3 - above). So a single-utterance call must not land on an engine
        # sized for 32, and a server that alternates between a wide batch and a long single utterance must not re-pack the weights and
        # re-capture the step graphs on every call. Each resident engine costs its own weight copies (~0.7-1.5 GB for Mini-v1).
        kv8_on = bool(getattr(self, "decoder_kv_fp8", False))
        if kv8_on and dt != torch.bfloat16:
            raise NotImplementedError("decoder_kv_fp8 needs the model in bfloat16 (e4m3 cache rows, bf16 activations)")
        kv8 = kv8_on and B > 8  # the GEMV step of up to 8 utterances keeps the bf16 cache
        key = (dev, dt, (fp8, kv8_on), "b<=4" if B <= 4 else ("b<=8" if B <= 8 else "b>8"))
        engines = self.__dict__.setdefault("_engines", {})
        e = engines.get(key)
        if e is None or e.cfg.max_batch < B or e.cfg.max_enc < N or e.cfg.max_prompt < P + 1 + T or e.cfg.max_ctx < P + max_length:
            caps = dict(max_batch=B, max_ctx=max(P + max_length, 64), max_enc=max(N, 16), max_prompt=max(P + 1 + T, 8))
            if e is not None:
                caps = {k: max(v, int(getattr(e.cfg, k))) for k, v in caps.items()}
                engines.pop(key).close()  # gone from the cache before the new one is built: a failed creation must not leave a closed engine behind
            for k in [k for k in engines if k[:3] != key[:3]]:  # the model moved / changed dtype / weight format: stale engines go
                engines.pop(k).close()
            d = self.config.decoder
            e = DecoderEngine(hidden_size=d.hidden_size, num_layers=d.num_hidden_layers, num_heads=d.num_attention_heads, ffn_dim=d.ffn_dim,
                              num_codebooks=d.num_codebooks, vocab_size=d.vocab_size, max_positions=d.max_position_embeddings,
                              rope=d.rope_embeddings, rope_theta=d.rope_theta, pad_token_id=d.pad_token_id, eos_token_id=d.eos_token_id,
                              bos_token_id=d.bos_token_id, dtype=dt, device=dev, num_kv_heads=d.num_key_value_heads,
                              num_cross_kv_heads=d.num_cross_attention_key_value_heads, weights_fp8=fp8, kv_fp8=kv8, **caps)
            e.load_state_dict(self.decoder.state_dict())
            engines[key] = e
        self.__dict__["_engine_last"] = e
        return e

    @property
    def _engine(self) -> Optional[DecoderEngine]:
        """The decoder engine the last call ran on (tests read its ids); assigning ``None`` drops every cached engine (weights, device or
        dtype changed: they re-pack lazily)."""
        return self.__dict__.get("_engine_last")

    @_engine.setter
    def _engine(self, value):
        if value is None:
            self.__dict__["_engines"] = {}
            t5 = self.__dict__.pop("_t5_engine", None)  # the description encoder's packed weight copy goes with them
            if t5 is not None:
                t5.close()
        self.__dict__["_engine_last"] = value

    # -- the pieces of generate() that stay on the torch side (once per call, off the per-token loop) ---------------
    def _encode_description_eager(self, input_ids, attention_mask):
        enc = self.text_encoder(input_ids=input_ids, attention_mask=attention_mask, return_dict=True).last_hidden_state
        if hasattr(self, "enc_to_dec_proj"):
            enc = self.enc_to_dec_proj(enc)
        if attention_mask is not None:
            enc = enc * attention_mask[..., None]
        return enc

    def _get_t5_engine(self, B: int, N: int) -> Optional[T5Engine]:
        """The HIP description encoder for this model's text encoder, or None when the text encoder is not a T5 the engine implements
        (then the stock module runs, as SURVEY.md section 2 row 8 scopes it) or ``use_native_text_encoder`` is off (A/B, parity tests).
        Capacities grow-only; the packed weight copy (0.68 GB for flan-t5-large in bf16) is dropped with the decoder engines."""
        if not getattr(self, "use_native_text_encoder", True) or os.environ.get("PTTS_NO_NATIVE_T5", "0") not in ("", "0"):
            return None
        cfg = getattr(self.text_encoder, "config", None)
        if cfg is None or not T5Engine.supports(cfg) or self.dtype not in (torch.float32, torch.bfloat16):
            return None
        e = self.__dict__.get("_t5_engine")
        if e is None or e.device != self.device or e.dtype != self.dtype or e.max_batch < B or e.max_len < N:
            caps = dict(max_batch=B, max_len=max(N, 16))
            if e is not None:
                caps = dict(max_batch=max(B, e.max_batch), max_len=max(caps["max_len"], e.max_len))
                self.__dict__.pop("_t5_engine").close()
            e = T5Engine.from_config(cfg, dtype=self.dtype, device=self.device, **caps)
            e.load_state_dict(self.text_encoder.state_dict())
            self.__dict__["_t5_engine"] = e
        return e

    def _encode_description(self, input_ids, attention_mask):
        """:3048-3097 — T5 encoder, optional enc_to_dec_proj, masked positions zeroed. On a HIP device the encoder is the native one
        (``ptts_t5_encode``: 7 kernel nodes per block in one captured hipGraph). A text encoder the HIP
        engine does not implement runs as the stock module (~360 tiny launches, 13 ms eager at 64 tokens), captured once per
        (batch, length, masked?) into a torch HIP graph and replayed (~3-5 ms); a capture failure there falls back to the eager call."""
        if input_ids.device.type == "cuda":
            t5 = self._get_t5_engine(int(input_ids.shape[0]), int(input_ids.shape[1]))
            if t5 is not None:
                enc = t5.encode(input_ids, attention_mask)  # fp32, masked positions already zero
                if hasattr(self, "enc_to_dec_proj"):
                    enc = self.enc_to_dec_proj(enc.to(self.dtype))
                    if attention_mask is not None:
                        enc = enc * attention_mask[..., None]
                return enc
        if input_ids.device.type != "cuda" or not getattr(self, "use_encoder_graph", True):
            return self._encode_description_eager(input_ids, attention_mask)
        key = (tuple(input_ids.shape), attention_mask is not None, self.dtype, input_ids.device)
        cache = self.__dict__.setdefault("_enc_graphs", {})
        entry = cache.get(key)
        if entry is None:
            try:
                static_ids = input_ids.clone()
                static_mask = attention_mask.clone() if attention_mask is not None else None
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(2):  # warm-up outside capture (lazy inits, autotuning)
                        self._encode_description_eager(static_ids, static_mask)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = self._encode_description_eager(static_ids, static_mask)
                entry = (graph, static_ids, static_mask, static_out)
            except Exception:  # noqa: BLE001 — capture is an optimisation only
                entry = False
            if len(cache) > 16:
                cache.clear()
            cache[key] = entry
        if entry is False:
            return self._encode_description_eager(input_ids, attention_mask)
        graph, static_ids, static_mask, static_out = entry
        static_ids.copy_(input_ids)
        if static_mask is not None:
            static_mask.copy_(attention_mask)
        graph.replay()
        return static_out.clone()

    @torch.no_grad()
    def generate(self, inputs: Optional[torch.Tensor] = None, generation_config=None, logits_processor=None, stopping_criteria=None,
                 synced_gpus: Optional[bool] = None, streamer=None, **kwargs):
        """Same call surface as the reference (:3321-3653). Returns the waveform [batch, samples] (zero padded), or a
        ``GenerateOutput`` with ``.sequences`` = waveform and ``["audios_length"]`` when ``return_dict_in_generate``."""
        gc = copy.deepcopy(generation_config if generation_config is not None else self.generation_config)
        mk = gc.update(**kwargs)
        input_ids = inputs if inputs is not None else mk.pop("input_ids", None)
        attention_mask = mk.pop("attention_mask", None)
        prompt_input_ids = mk.pop("prompt_input_ids", None)
        prompt_attention_mask = mk.pop("prompt_attention_mask", None)
        prompt_hidden_states = mk.pop("prompt_hidden_states", None)
        encoder_outputs = mk.pop("encoder_outputs", None)
        input_values = mk.pop("input_values", None)
        decoder_input_ids = mk.pop("decoder_input_ids", None)
        check_model_kwargs(mk)  # unknown names raise like transformers' _validate_model_kwargs; forward arguments this path cannot honour raise too
        if mk.get("past_key_values") is not None and getattr(gc, "cache_implementation", None) is not None:
            raise ValueError("Passing both `cache_implementation` (used to initialize certain caches) and `past_key_values` (a "
                             "Cache object) is unsupported. Please use only one of the two.")
        if getattr(gc, "cache_implementation", None) == "quantized":
            raise ValueError("This model does not support the quantized cache. If you want your model to support quantized "
                             "cache, please open an issue on the Parler-TTS repository https://github.com/huggingface/parler-tts")
        if getattr(gc, "cache_implementation", None) == "sliding_window":
            # the reference clamps the self-attention cache to `config.sliding_window` (:3269-3270; a field its own config classes do not
            # define, so the stock model raises AttributeError there). A windowed cache changes the arithmetic; the HIP engine's static
            # arena does not implement it, and nothing is silently ignored.
            raise NotImplementedError("cache_implementation='sliding_window' is not implemented by the HIP engine (its KV arena is a full static "
                                      "cache: use cache_implementation=None or 'static')")
        check_generation_mode(gc)  # greedy / sampling only (:3574-3578)
        dev = self.device
        d = self.config.decoder
        K = d.num_codebooks
        # --- description / prompt conditioning ---------------------------------------------------------------------
        if encoder_outputs is not None:
            enc = encoder_outputs[0] if isinstance(encoder_outputs, (tuple, list)) else encoder_outputs.last_hidden_state
        else:
            if input_ids is None:
                raise ValueError("`input_ids` (the tokenized description) or `encoder_outputs` must be given")
            input_ids = input_ids.to(dev)
            enc = self._encode_description(input_ids, attention_mask.to(dev) if attention_mask is not None else None)
        B = enc.shape[0]
        if streamer is not None and B > 1:
            raise ValueError("ParlerTTSStreamer only supports batch size 1")
        enc = enc.to(dev).float()
        enc_mask = attention_mask.to(dev) if attention_mask is not None else None
        prompt, prompt_mask = None, None
        if prompt_hidden_states is None and prompt_input_ids is not None:
            prompt_hidden_states = self.embed_prompts(prompt_input_ids.to(dev))  # :3100
        if prompt_hidden_states is not None:
            ph = prompt_hidden_states.to(dev).float()
            pm = prompt_attention_mask.to(dev) if prompt_attention_mask is not None else None
            if self.prompt_cross_attention:  # :3102-3128: prompt joins the cross-attention context
                ph = ph + self.embed_positions.weights[: ph.shape[1]].float()[None]
                if pm is not None and enc_mask is None:
                    enc_mask = torch.ones(enc.shape[:2], device=dev, dtype=pm.dtype)
                elif enc_mask is not None and pm is None:
                    pm = torch.ones(ph.shape[:2], device=dev, dtype=enc_mask.dtype)
                enc = torch.cat([enc, ph], dim=1)
                if pm is not None:
                    enc_mask = torch.cat([enc_mask, pm], dim=1)
            else:
                prompt, prompt_mask = ph, pm
        num_return = int(getattr(gc, "num_return_sequences", 1) or 1)
        if num_return > 1:  # _expand_inputs_for_generation (:3556-3561)
            enc = enc.repeat_interleave(num_return, 0)
            enc_mask = enc_mask.repeat_interleave(num_return, 0) if enc_mask is not None else None
            prompt = prompt.repeat_interleave(num_return, 0) if prompt is not None else None
            prompt_mask = prompt_mask.repeat_interleave(num_return, 0) if prompt_mask is not None else None
            B *= num_return
        N = enc.shape[1]
        P = prompt.shape[1] if prompt is not None else 0
        bos = d.bos_token_id
        # --- voice prompt (:3136-3194): `input_values` -> DAC codes -> `decoder_input_ids`, continued by the decoder --------------
        if input_values is not None:
            enc_out = self.audio_encoder.encode(input_values.to(dev), n_quantizers=K)
            audio_codes = enc_out.audio_codes
            if audio_codes.dim() == 4:
                if audio_codes.shape[0] != 1:  # :3182-3186
                    raise ValueError(f"Expected 1 frame in the audio code outputs, got {audio_codes.shape[0]} frames. Ensure chunking is "
                                     "disabled by setting `chunk_length=None` in the audio encoder.")
                audio_codes = audio_codes[0]
            decoder_input_ids = audio_codes.reshape(audio_codes.shape[0] * K, audio_codes.shape[-1])
        prefix = None
        if decoder_input_ids is not None:
            decoder_input_ids = decoder_input_ids.to(dev).long()
            if decoder_input_ids.dim() != 2 or decoder_input_ids.shape[0] % K:
                raise ValueError(f"decoder_input_ids must be [batch * num_codebooks, frames], got {tuple(decoder_input_ids.shape)}")
            if not bool((decoder_input_ids[..., 0] != bos).all()):  # already starts with the BOS column (:3017-3018)
                decoder_input_ids = decoder_input_ids[:, 1:]
            if decoder_input_ids.shape[0] // K == B // num_return and num_return > 1:
                decoder_input_ids = decoder_input_ids.reshape(-1, K, decoder_input_ids.shape[-1]).repeat_interleave(num_return, 0).reshape(B * K, -1)
            if decoder_input_ids.shape[0] != B * K:
                raise ValueError(f"decoder_input_ids batch {decoder_input_ids.shape[0] // K} != batch {B}")
            if decoder_input_ids.shape[-1] > 0:
                prefix = decoder_input_ids
        T0 = prefix.shape[-1] if prefix is not None else 0
        # --- lengths (:3458-3469): counted from the 1 + T0 given decoder columns ---------------------------------------------------
        if gc.max_new_tokens is not None:
            max_length = int(gc.max_new_tokens) + 1 + T0
        else:
            max_length = int(gc.max_length)
        gc.max_length = max_length  # what _prepare_generated_length leaves in the config (ForcedEOS / length-penalty processors read it)
        min_new = int(gc.min_new_tokens or 0)
        if getattr(gc, "min_length", 0):
            min_new = max(min_new, int(gc.min_length) - 1 - T0)
        if max_length < T0 + 2:
            raise ValueError(f"Input length of decoder_input_ids is {T0 + 1}, but `max_length` is set to {max_length}: no room for a generated token")
        if max_length < 2:
            raise ValueError("`max_length` / `max_new_tokens` leave no room for a generated token")
        do_sample = bool(gc.do_sample)
        manual = (logits_processor is not None and len(logits_processor) > 0) or (stopping_criteria is not None and len(stopping_criteria) > 0)
        # GenerationConfig options the device sampler does not implement (repetition / n-gram penalties, bad words, min-p, typical-p,
        # max_time ...): the reference honours them through transformers' _get_logits_processor (:3540-3552); here they run the host
        # loop with transformers' own processor objects (generation_extras.py). Nothing is silently ignored.
        extras = active_extras(gc)
        manual = manual or bool(extras)
        # return_dict_in_generate + output_scores / output_logits: the reference returns `_sample`'s per-step tuples next to the waveform
        # (:3648-3651 keeps the ModelOutput); they exist on the host loop only
        as_dict = bool(getattr(gc, "return_dict_in_generate", False))
        keep_scores = as_dict and bool(getattr(gc, "output_scores", False))
        keep_logits = as_dict and bool(getattr(gc, "output_logits", False))
        if as_dict and (getattr(gc, "output_attentions", False) or getattr(gc, "output_hidden_states", False)):
            raise NotImplementedError("output_attentions / output_hidden_states: the HIP decoder does not materialise attention maps or per-layer states")
        manual = manual or keep_scores or keep_logits
        self._step_records = ([] if keep_scores else None, [] if keep_logits else None)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if do_sample else 0  # follows torch.manual_seed()
        gen_kw = dict(max_length=max_length, min_new_tokens=min_new, do_sample=do_sample, temperature=float(gc.temperature or 1.0),
                      top_k=int(gc.top_k or 0) if do_sample else 0, top_p=float(gc.top_p if gc.top_p is not None else 1.0),
                      use_eos_gate=logits_processor is None)
        # (one engine per call: n independent sub-batches on n engines and streams - with or without disjoint CU masks - never beat it on one GPU:
        #  profiles/r04_experiments.txt calls 20 / 26, profiles/r06_experiments.txt call 12; the split loop is gone since round 6)
        eng = self._get_engine(B, N, P, max_length, T0)
        eng.set_gen_params(seed=seed, **gen_kw)
        pad, eos = gc.pad_token_id if gc.pad_token_id is not None else d.pad_token_id, d.eos_token_id
        bos_col = torch.full((B * K, 1), bos, dtype=torch.long, device=dev)
        dec_ids = bos_col if prefix is None else torch.cat([bos_col, prefix], dim=-1)  # :3011-3018
        delayed, pattern = build_delay_pattern_mask(dec_ids, bos, pad, max_length, K)  # :3523-3530
        if streamer is not None:
            streamer.put(delayed.cpu())  # :3533-3534
        eng.set_audio_prefix(prefix)
        wav_pre = None
        # opt-in (model.overlap_codec = True): on ONE GPU the overlap does not pay (bs=32: 1513 -> 1575 ms per generate(), the codec's
        # MFMA kernels take CUs from the latency-bound token kernels; bs=1: the per-chunk polling costs what the 5.7 ms decode saves)
        if not manual and streamer is None and dev.type == "cuda" and getattr(self, "overlap_codec", False) and max_length - K >= 96:
            output_ids, wav_pre = self._run_device_loop_overlap(eng, enc, enc_mask, prompt, prompt_mask, max_length, delayed.shape[1])
        elif not manual:
            output_ids = self._run_device_loop(eng, enc, enc_mask, prompt, prompt_mask, max_length, streamer, delayed.shape[1], min_new)
        else:
            hf_list = None
            if extras:
                custom = logits_processor if logits_processor is not None else [ParlerTTSLogitsProcessor(eos, K, B, dev)]  # :3418
                enc_ids = input_ids if (input_ids is not None and torch.is_tensor(input_ids) and input_ids.dim() == 2) else None
                hf_list = build_processors(gc, delayed.shape[1], enc_ids, custom, dev, eos)
                cfg_criteria, user_criteria = build_stopping_criteria(gc, stopping_criteria)
                stopping_criteria = cfg_criteria + user_criteria
            output_ids = self._run_host_loop(eng, enc, enc_mask, prompt, prompt_mask, max_length, min_new, gc, logits_processor,
                                             stopping_criteria, streamer, eos, pad, delayed, hf_list=hf_list)
        if streamer is not None:
            streamer.end()
        wav, lengths = self._undelay_and_decode(output_ids, pattern, bos_col, bos, pad, wav_pre)
        if as_dict:
            out = GenerateOutput(sequences=wav, audios_length=lengths)
            if keep_scores:
                out["scores"] = tuple(self._step_records[0])
            if keep_logits:
                out["logits"] = tuple(self._step_records[1])
            self._step_records = (None, None)
            return out
        return wav

    def _undelay_and_decode(self, output_ids, pattern, bos_col, bos, pad, wav_pre=None):
        """The tail of generate() after the token loop: un-delay (:3585-3597), per-sample special-id filter and codec (:3600-3647).
        Returns (waveform [B, samples] zero-padded, per-sample lengths)."""
        d = self.config.decoder
        K, dev = d.num_codebooks, output_ids.device
        B = output_ids.shape[0] // K
        # The reference rebuilds the mask from the un-delayed `input_ids` (:3589-3594); only its BOS / PAD triangles are
        # tested (:3596), and audio codes are never BOS / PAD, so the BOS column alone gives the identical keep-mask.
        output_ids = apply_delay_pattern_mask(output_ids, pattern)
        _, m2 = build_delay_pattern_mask(bos_col, bos, pad, output_ids.shape[1], K)
        keep = (m2 != bos) & (m2 != pad)
        codes = output_ids[keep].reshape(B, K, -1)
        cb = self.audio_encoder.config.codebook_size
        bad = codes >= cb
        if not bool(bad.any()):
            if wav_pre is not None and wav_pre.shape == (B, codes.shape[-1] * self._codec_hop()):
                wav = wav_pre  # decoded chunk by chunk on the side stream while the token loop ran (same samples as one full decode)
            else:
                wav = self.audio_encoder.decode(audio_codes=codes[None], audio_scales=[None] * B).audio_values.squeeze(1)
            lengths = [int(wav.shape[1])] * B
        elif callable(getattr(self.audio_encoder, "decode_filtered", None)):
            # per-sample branch (:3627-3647: drop every column holding a special id, decode, zero-pad) as ONE filter kernel + ONE ragged
            # codec pass over the whole batch; the only host synchronisation is the read of the B kept-frame counts for `audios_length`
            wav_full, frames = self.audio_encoder.decode_filtered(codes[None])
            hop = wav_full.shape[-1] // max(codes.shape[-1], 1)
            nf = [int(x) for x in frames.tolist()]
            lengths = [n * hop if n > 0 else 1 for n in nf]  # an utterance with no valid frame yields torch.zeros(1) in the reference (:3641)
            wav = wav_full[:, 0, :max(lengths)]
        else:  # a codec without the ragged entry points (any AutoModel-registered codec: SURVEY §8(b)): the reference's loop, literally
            outs: List[torch.Tensor] = []
            for b in range(B):
                ok = bad[b].sum(dim=0) == 0
                if int(ok.sum()) > 0:
                    w = self.audio_encoder.decode(audio_codes=codes[b: b + 1, :, ok][None], audio_scales=[None]).audio_values[0, 0]
                else:
                    w = torch.zeros(1, device=dev)
                outs.append(w)
            lengths = [int(w.shape[0]) for w in outs]
            wav = torch.nn.utils.rnn.pad_sequence(outs, batch_first=True, padding_value=0)
        return wav, lengths

    # -- default path: the whole `_sample` loop runs on the device -------------------------------------------------------
    def _run_device_loop(self, eng, enc, enc_mask, prompt, prompt_mask, max_length, streamer, given: int = 1, min_new: int = 0):
        eng.prefill(enc, enc_mask, prompt, prompt_mask, sample=True)  # BOS column + voice-prompt columns, then the first sampled one
        remaining = max_length - given - 1
        if streamer is None:
            done, made = False, 1  # columns generated so far (the prefill's tail produced the first)
            while not done and remaining > 0:
                n = min(64, remaining)
                eng.decode_steps(n)
                remaining -= n
                made += n
                # no row can finish while EOS is still blocked by min_new_tokens (and max_length is not reached): the host does not
                # synchronise with the device inside that stretch, the graph launches of the next chunk queue up behind this one
                if made > min_new or remaining == 0:
                    _, done = eng.state()
            return eng.ids()
        # Streaming: the decoder graph runs AHEAD on the main stream while the finished columns of the previous chunk are
        # forwarded to the streamer on a side stream (one put per column, like `_sample`; the streamer's un-delay + chunked DAC
        # decode is enqueued on that side stream too), so the codec never stalls the token loop. Only the columns the COMPLETED
        # boundary event covers are forwarded (given + 1 + steps enqueued up to that boundary): steps still in flight may already
        # have bumped `cur_len` while their ids column is not visible to a concurrent copy yet (the tail kernel orders the two
        # stores at workgroup scope only). Steps after every row finished are device-side no-ops.
        chunk = int(getattr(streamer, "play_steps", 16) or 16)
        if self.device.type != "cuda":  # host-logic tests drive this loop with a stand-in engine: same protocol, no streams
            sent, done = given, False
            while True:
                ids = eng.ids()
                for j in range(sent, ids.shape[1]):
                    streamer.put(ids[:, j].cpu())
                sent = ids.shape[1]
                if done or remaining <= 0:
                    return eng.ids()
                n = min(chunk, remaining)
                eng.decode_steps(n)
                remaining -= n
                _, done = eng.state()
        main = torch.cuda.current_stream(self.device)
        side = self.__dict__.get("_stream_side")
        if side is None or side.device != self.device:
            side = self.__dict__["_stream_side"] = torch.cuda.Stream(self.device, priority=-1)  # its own (high-priority) hardware queue
        sent = given
        first = min(max(chunk - given - 1, 0), remaining)  # land exactly on the streamer's first `play_steps` boundary
        pump = _StepPump(eng, main, first, chunk, remaining)
        while pump.wait_boundary():      # the oldest chunk in flight has finished; later steps keep the GPU busy meanwhile
            with torch.cuda.stream(side):
                cols = eng.ids(max_cols=given + 1 + pump.done_steps)[:, sent:].cpu()  # one copy for the chunk, then one put per column like `_sample`
                for j in range(cols.shape[1]):
                    streamer.put(cols[:, j])
                sent += cols.shape[1]
                _, done = eng.state()
            if done:
                pump.stop()
        main.synchronize()
        with torch.cuda.stream(side):
            cols = eng.ids()[:, sent:].cpu()
            for j in range(cols.shape[1]):
                streamer.put(cols[:, j])
        side.synchronize()
        return eng.ids()

    def _codec_hop(self) -> int:
        return int(math.prod(getattr(self.audio_encoder, "decoder_rates", (8, 8, 4, 2))))

    def _run_device_loop_overlap(self, eng, enc, enc_mask, prompt, prompt_mask, max_length, given: int = 1):
        """The default (non-streaming) loop with the codec overlapped (BASELINE configs[4] "streaming DAC decode"): the token
        graph runs one chunk ahead on the main stream; on a side stream the frames completed by the previous chunk are
        un-delayed and decoded with ``ptts_dac_decode_chunk`` straight into the final waveform, keeping one receptive field
        of frames back as the right halo, so the samples are those of ONE full-utterance decode. Falls back (returns no
        waveform) as soon as a special id (EOS / padding >= codebook_size) shows up: those runs need the reference's
        per-sample column filter (:3627-3647) before decoding."""
        from .streamer import receptive_halo_frames

        dev = self.device
        K = self.config.decoder.num_codebooks
        ae = self.audio_encoder
        hop, halo, cb = self._codec_hop(), receptive_halo_frames(getattr(ae, "decoder_rates", (8, 8, 4, 2))), ae.config.codebook_size
        B, F = enc.shape[0], max_length - K
        eng.prefill(enc, enc_mask, prompt, prompt_mask, sample=True)
        main = torch.cuda.current_stream(dev)
        side = self.__dict__.get("_stream_side")
        if side is None or side.device != dev:
            side = self.__dict__["_stream_side"] = torch.cuda.Stream(dev, priority=-1)
        wav = torch.empty(B, F * hop, dtype=torch.float32, device=dev)
        codes = torch.zeros(B, K, F, dtype=torch.long, device=dev)
        wav.record_stream(side)
        codes.record_stream(side)
        side.wait_stream(main)
        have = emitted = 0
        ok, done = True, False

        def absorb(final: bool):
            nonlocal have, emitted, ok, done
            cur, done = eng.state()
            if not ok:
                return
            ids = eng.ids().view(B, K, -1)
            f_new = max(0, min(ids.shape[-1] - K, F))
            if f_new > have:
                for k in range(K):  # un-delay: the code of (codebook k, frame f) sits in column f + 1 + k
                    codes[:, k, have:f_new] = ids[:, k, have + 1 + k: f_new + 1 + k]
                if bool((codes[:, :, have:f_new] >= cb).any()):
                    ok = False
                    return
                have = f_new
            keep_back = 0 if (final and have == F) else halo
            if have - keep_back > emitted:
                ae.decode_chunk(codes[None], emitted, have - emitted, halo, out=wav, n_emit=have - keep_back - emitted)
                emitted = have - keep_back

        remaining = max_length - given - 1
        pump = _StepPump(eng, main, min(64, remaining), 64, remaining, max_ahead=64)  # one chunk ahead of the codec
        while pump.wait_boundary():
            with torch.cuda.stream(side):
                absorb(final=False)
            if done:
                pump.stop()
        main.synchronize()
        with torch.cuda.stream(side):
            absorb(final=True)
        side.synchronize()
        ids = eng.ids()
        return ids, (wav if (ok and emitted == F and ids.shape[1] == max_length) else None)

    # -- user LogitsProcessorList / StoppingCriteria: forward on the HIP engine, selection in torch --------------------------
    def _run_host_loop(self, eng, enc, enc_mask, prompt, prompt_mask, max_length, min_new, gc, processors, criteria, streamer, eos, pad,
                       given_ids, hf_list=None):
        """``hf_list``: the COMPLETE processor list of the call (config processors, custom / default list, warpers) built by
        generation_extras.build_processors; when given it replaces the inline MinNewTokens / warper code below."""
        dev = self.device
        eng.prefill(enc, enc_mask, prompt, prompt_mask, sample=False)
        B = enc.shape[0]
        K = self.config.decoder.num_codebooks
        seq = given_ids.clone()  # BOS column (+ the delayed voice-prompt columns)
        given = seq.shape[-1]
        unfinished = torch.ones(B * K, dtype=torch.long, device=dev)
        if processors is None:  # :3418: the default LogitsProcessorList([ParlerTTSLogitsProcessor]) whenever none is passed,
            processors = [ParlerTTSLogitsProcessor(eos, K, B, dev)]  # with or without user stopping criteria
        while True:
            scores = eng.logits().float()
            rec_scores, rec_logits = getattr(self, "_step_records", (None, None))
            if rec_logits is not None:
                rec_logits.append(scores.clone())
            if hf_list is not None:  # transformers' own processor objects in its order: config -> custom / default list -> warpers
                scores = hf_list(seq, scores)
            else:
                if min_new > 0 and (seq.shape[-1] - given) < min_new:
                    scores[:, eos] = -math.inf
                for proc in (processors or []):
                    scores = proc(seq, scores)
                if gc.do_sample:
                    if gc.temperature and gc.temperature != 1.0:
                        scores = scores / gc.temperature
                    if gc.top_k:
                        kth = torch.topk(scores, min(int(gc.top_k), scores.shape[-1]))[0][..., -1, None]
                        scores = scores.masked_fill(scores < kth, -math.inf)
                    if gc.top_p is not None and gc.top_p < 1.0:
                        sl, si = torch.sort(scores, descending=False)
                        rm = sl.softmax(dim=-1).cumsum(dim=-1) <= (1 - gc.top_p)
                        rm[..., -1:] = False
                        scores = scores.masked_fill(rm.scatter(1, si, rm), -math.inf)
            if rec_scores is not None:
                rec_scores.append(scores)
            if gc.do_sample:
                nxt = torch.multinomial(torch.softmax(scores, dim=-1), 1).squeeze(1)
            else:
                nxt = torch.argmax(scores, dim=-1)
            nxt = nxt * unfinished + pad * (1 - unfinished)
            seq = torch.cat([seq, nxt[:, None]], dim=-1)
            done = (nxt == eos) | (seq.shape[-1] >= max_length)
            for crit in (criteria or []):
                r = crit(seq, scores)
                done = done | (r if torch.is_tensor(r) else torch.full_like(done, bool(r)))
            unfinished = unfinished & ~done.long()
            eng.push_tokens(nxt, (1 - unfinished).int())
            if streamer is not None:
                streamer.put(nxt.cpu())
            if int(unfinished.max()) == 0:
                break
            eng.step_forward()
        return seq
