"""Configuration objects with the reference's field names, defaults and JSON layout
(parler_tts/configuration_parler_tts.py:33-291, dac_wrapper/configuration_dac.py:7-27), so a released
checkpoint's ``config.json`` loads unchanged. They are plain, dependency-light classes (no coupling to a
particular transformers release: the reference pins 4.46.1, this image ships 5.x); only the third-party text
encoder config is materialised through ``transformers.AutoConfig``.
"""
from __future__ import annotations

import copy
import json
import os
from typing import Any, Dict, Optional


class _Config:
    model_type = ""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to_dict(self) -> Dict[str, Any]:
        out = {}
        for k, v in self.__dict__.items():
            if k.startswith("_"):
                continue
            out[k] = v.to_dict() if hasattr(v, "to_dict") else copy.deepcopy(v)
        out["model_type"] = self.model_type
        return out

    @classmethod
    def from_dict(cls, d: Dict[str, Any], **kwargs):
        d = dict(d)
        d.pop("model_type", None)
        d.update(kwargs)
        return cls(**d)

    @classmethod
    def from_pretrained(cls, path: str, **kwargs):
        f = path if os.path.isfile(path) else os.path.join(path, "config.json")
        with open(f) as fh:
            return cls.from_dict(json.load(fh), **kwargs)

    def to_json_string(self) -> str:
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def save_pretrained(self, save_directory: str):
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as fh:
            fh.write(self.to_json_string())

    def __repr__(self):
        return f"{self.__class__.__name__} {self.to_json_string()}"


class ParlerTTSDecoderConfig(_Config):
    """Fields and defaults of the reference decoder config (configuration_parler_tts.py:111-172)."""

    model_type = "parler_tts_decoder"
    keys_to_ignore_at_inference = ["past_key_values"]

    def __init__(self, vocab_size=2049, max_position_embeddings=2048, num_hidden_layers=24, ffn_dim=4096,
                 num_attention_heads=16, num_key_value_heads=None, num_cross_attention_key_value_heads=None, layerdrop=0.0,
                 use_cache=True, activation_function="gelu", hidden_size=1024, dropout=0.1, attention_dropout=0.0,
                 activation_dropout=0.0, initializer_factor=0.02, scale_embedding=False, num_codebooks=4, pad_token_id=2048,
                 bos_token_id=2049, eos_token_id=2048, tie_word_embeddings=False, rope_embeddings=False, rope_theta=10_000.0,
                 cross_attention_implementation_strategy=None, use_fused_lm_heads=False, codebook_weights=None, **kwargs):
        if codebook_weights is not None and len(codebook_weights) != num_codebooks:
            raise ValueError(f"`codebook_weights` has length {len(codebook_weights)} when it should be of length {num_codebooks}.")
        num_key_value_heads = num_attention_heads if num_key_value_heads is None else num_key_value_heads
        if num_cross_attention_key_value_heads is None:
            num_cross_attention_key_value_heads = num_key_value_heads
        super().__init__(
            vocab_size=vocab_size, max_position_embeddings=max_position_embeddings, hidden_size=hidden_size, ffn_dim=ffn_dim,
            num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads, num_key_value_heads=num_key_value_heads,
            num_cross_attention_key_value_heads=num_cross_attention_key_value_heads, dropout=dropout,
            attention_dropout=attention_dropout, activation_dropout=activation_dropout, activation_function=activation_function,
            initializer_factor=initializer_factor, layerdrop=layerdrop, use_cache=use_cache, scale_embedding=scale_embedding,
            num_codebooks=num_codebooks, rope_embeddings=rope_embeddings, rope_theta=rope_theta,
            cross_attention_implementation_strategy=cross_attention_implementation_strategy, use_fused_lm_heads=use_fused_lm_heads,
            codebook_weights=codebook_weights, pad_token_id=pad_token_id, bos_token_id=bos_token_id, eos_token_id=eos_token_id,
            tie_word_embeddings=tie_word_embeddings, **kwargs)

    def check_supported_by_engine(self):
        """The HIP engine implements the released Mini/Large-v1 architecture family; anything else fails loudly."""
        for n in (self.num_key_value_heads, self.num_cross_attention_key_value_heads):  # grouped-query attention (repeat_kv :280-289)
            if n < 1 or self.num_attention_heads % n:
                raise ValueError(f"num_attention_heads {self.num_attention_heads} must be divisible by the K/V head count {n}")
        if self.activation_function != "gelu":
            raise NotImplementedError(f"activation_function={self.activation_function!r}: only exact-erf 'gelu' is implemented by the HIP engine")
        if self.hidden_size // self.num_attention_heads != 64:
            raise NotImplementedError("the HIP engine implements head_dim == 64 (Mini-v1 / Large-v1)")
        if self.scale_embedding:
            pass  # the reference computes embed_scale but never applies it (modeling_parler_tts.py:1350, :1433)


class DACConfig(_Config):
    """dac_wrapper/configuration_dac.py:7-27 (model_type "dac_on_the_hub" for transformers > 4.44.2, "dac" before)."""

    model_type = "dac_on_the_hub"

    def __init__(self, num_codebooks: int = 9, model_bitrate: int = 8, codebook_size: int = 1024, latent_dim: int = 1024,
                 frame_rate: int = 86, sampling_rate: int = 44100, **kwargs):
        super().__init__(num_codebooks=num_codebooks, model_bitrate=model_bitrate, codebook_size=codebook_size, latent_dim=latent_dim,
                         frame_rate=frame_rate, sampling_rate=sampling_rate, **kwargs)


def _sub_config(d, default_type: Optional[str] = None):
    """dict | config object → config object. The text encoder is third-party: materialise it via AutoConfig."""
    if not isinstance(d, dict):
        return d
    d = dict(d)
    mt = d.pop("model_type", default_type)
    if mt in ("dac", "dac_on_the_hub"):
        return DACConfig(**d)
    if mt == "parler_tts_decoder":
        return ParlerTTSDecoderConfig(**d)
    from transformers import AutoConfig

    return AutoConfig.for_model(mt, **d)


class ParlerTTSConfig(_Config):
    """Composite config (configuration_parler_tts.py:175-291): text_encoder / audio_encoder / decoder sub-configs,
    ``vocab_size`` of the prompt tokenizer and ``prompt_cross_attention``."""

    model_type = "parler_tts"
    is_composition = True

    def __init__(self, vocab_size=1024, prompt_cross_attention=False, **kwargs):
        if "text_encoder" not in kwargs or "audio_encoder" not in kwargs or "decoder" not in kwargs:
            raise ValueError("Config has to be initialized with text_encoder, audio_encoder and decoder config")
        text_encoder = _sub_config(kwargs.pop("text_encoder"))
        audio_encoder = _sub_config(kwargs.pop("audio_encoder"), "dac_on_the_hub")
        decoder = _sub_config(kwargs.pop("decoder"), "parler_tts_decoder")
        if isinstance(decoder, dict) or not isinstance(decoder, ParlerTTSDecoderConfig):
            decoder = ParlerTTSDecoderConfig(**(decoder if isinstance(decoder, dict) else decoder.to_dict()))
        kwargs["is_encoder_decoder"] = True
        super().__init__(vocab_size=vocab_size, prompt_cross_attention=prompt_cross_attention, text_encoder=text_encoder,
                         audio_encoder=audio_encoder, decoder=decoder, **kwargs)

    @classmethod
    def from_sub_models_config(cls, text_encoder_config, audio_encoder_config, decoder_config, **kwargs):
        return cls(text_encoder=text_encoder_config.to_dict(), audio_encoder=audio_encoder_config.to_dict(),
                   decoder=decoder_config.to_dict(), **kwargs)

    @property
    def sampling_rate(self):
        return self.audio_encoder.sampling_rate
