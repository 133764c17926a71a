"""``DACModel`` with the reference wrapper's interface (parler_tts/dac_wrapper/modeling_dac.py:13-164): same config
fields, same ``decode(audio_codes, audio_scales, padding_mask=None, return_dict=None)`` signature and the same
"one frame only" error, but ``decode`` and ``encode`` (voice-prompt input, :33-104) run on the HIP DAC engine
(csrc/ptts_dac.hip) instead of descript-audio-codec.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch

from ..configuration_parler_tts import DACConfig
from ..engine import DacEngine, fold_weight_norm


@dataclass
class DACEncoderOutput:
    """Stands in for transformers' EncodecEncoderOutput: ``.audio_codes`` [chunks=1, batch, codebooks, frames], ``.audio_scales``."""

    audio_codes: Optional[torch.Tensor] = None
    audio_scales: Optional[list] = None

    def __getitem__(self, i):
        return (self.audio_codes, self.audio_scales)[i]

    def get(self, k, default=None):
        return getattr(self, k, default)


@dataclass
class DACDecoderOutput:
    """Stands in for transformers' EncodecDecoderOutput: ``.audio_values`` [batch, channels, samples]."""

    audio_values: Optional[torch.Tensor] = None

    def __getitem__(self, i):
        return (self.audio_values,)[i]


class DACModel(torch.nn.Module):
    config_class = DACConfig
    main_input_name = "input_values"

    # descript 44 kHz architecture constants behind dac.model.DAC(n_codebooks, latent_dim, codebook_size) (:24-28)
    CODEBOOK_DIM = 8
    DECODER_DIM = 1536
    DECODER_RATES = (8, 8, 4, 2)
    ENCODER_DIM = 64
    MAX_GROWN_FRAME_UTTERANCES = 32 * 900  # grow-only engine capacities up to this batch x frames product (~30 GB of activations at 44.1 kHz)

    def __init__(self, config: DACConfig, decoder_dim: Optional[int] = None, decoder_rates=None, codebook_dim: Optional[int] = None):
        super().__init__()
        self.config = config
        self.decoder_dim = decoder_dim or getattr(config, "decoder_dim", self.DECODER_DIM)
        self.decoder_rates = tuple(decoder_rates or getattr(config, "decoder_rates", self.DECODER_RATES))
        self.codebook_dim = codebook_dim or getattr(config, "codebook_dim", self.CODEBOOK_DIM)
        self.encoder_dim = getattr(config, "encoder_dim", self.ENCODER_DIM)
        self._weights: Dict[str, torch.Tensor] = {}  # dac.model.DAC names under the reference's "model." prefix
        self._engine: Optional[DacEngine] = None
        self._engine_dev = None
        self._dummy = torch.nn.Parameter(torch.zeros(1), requires_grad=False)  # tracks .to(device)

    @property
    def device(self):
        return self._dummy.device

    # -- weights: keep the reference's key names ("model.quantizer...", "model.decoder...") ----------------------
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        w = {k: v.detach() for k, v in state_dict.items() if k.startswith("model.")}
        if strict and not any(k.startswith("model.decoder.") for k in w):
            raise RuntimeError("DACModel.load_state_dict: no 'model.decoder.*' tensors found")
        self._weights = w
        self._engine = None
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def state_dict(self, *args, prefix: str = "", **kwargs):
        return {prefix + k: v for k, v in self._weights.items()}

    def save_pretrained(self, save_directory: str, safe_serialization: bool = True, **kwargs):
        """config.json + model.safetensors under the reference wrapper's key names (``model.*``; helpers/push_to_hub_scripts/push_dac_to_hub.py:19-26)."""
        import os

        from safetensors.torch import save_file

        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        save_file({k: v.detach().cpu().contiguous() for k, v in self._weights.items()}, os.path.join(save_directory, "model.safetensors"),
                  metadata={"format": "pt"})

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, *model_args, config: Optional[DACConfig] = None, **kwargs):
        import os

        from safetensors.torch import load_file

        path = pretrained_model_name_or_path
        if not os.path.isdir(path):
            from huggingface_hub import snapshot_download  # needs network / a local HF cache

            path = snapshot_download(path, allow_patterns=["*.json", "*.safetensors"])
        own = {k: kwargs[k] for k in ("decoder_dim", "decoder_rates", "codebook_dim") if k in kwargs}  # the rest are hub / Auto* kwargs
        m = cls(config or DACConfig.from_pretrained(path), **own)
        m.load_state_dict(load_file(os.path.join(path, "model.safetensors")))
        return m

    def _get_engine(self, batch: int, frames: int, need_encoder: bool = False, whole_batch: bool = False) -> DacEngine:
        """An engine that decodes `frames` frames of up to `batch` utterances per pass. decode() / encode() loop over sub-batches of the
        engine's ``max_batch``, so the activation buffers (~0.8 MB per frame-utterance at 44.1 kHz: 32 x 2580 frames would be 65 GB) are
        bounded by sizing the sub-batch, not the request; ``whole_batch`` (chunked decode into one output buffer) needs them all at once."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("DACModel runs on the HIP engine only: move the model to a cuda device (no CPU fallback)")
        e = self._engine
        # `.to(dtype=torch.bfloat16)` (INFERENCE.md:29-32 casts the whole model): bf16 MFMA operands with fp32 accumulation -
        # at least the precision of the reference's bf16 codec; float32 is the exact-f32 parity mode
        ok32 = self.config.latent_dim % 32 == 0 and self.decoder_dim % (32 << len(self.decoder_rates)) == 0  # 16x16x32 MFMA steps
        compute = torch.bfloat16 if (self._dummy.dtype == torch.bfloat16 and ok32) else torch.float32
        nf = max(frames, 64)
        nb = max(batch, 1) if whole_batch else min(max(batch, 1), max(1, self.MAX_GROWN_FRAME_UTTERANCES // nf))
        if (e is None or self._engine_dev != dev or e.max_batch < nb or e.max_frames < frames or (need_encoder and e.encoder_dim <= 0)
                or e.compute_dtype != compute):
            if e is not None:
                e.close()
            if not self._weights:
                raise RuntimeError("DACModel has no weights loaded")
            c = self.config
            has_enc = any(k.startswith("model.encoder.") for k in self._weights)
            if need_encoder and not has_enc:
                raise RuntimeError("DACModel.encode: the checkpoint holds no 'model.encoder.*' tensors")
            keep_enc = need_encoder or (e is not None and e.encoder_dim > 0)  # the encoder is built on first use only
            if e is not None and self._engine_dev == dev and e.compute_dtype == compute:
                # capacities grow with the calls (a wide batch after a long utterance keeps room for both: no re-packing when a server
                # alternates) as long as the activation buffers stay moderate: they scale with batch x frames, ~1 MB per frame-utterance
                gb, gf = max(nb, e.max_batch), max(nf, e.max_frames)
                if gb * gf <= self.MAX_GROWN_FRAME_UTTERANCES:
                    nb, nf = gb, gf
            e = DacEngine(num_codebooks=c.num_codebooks, codebook_size=c.codebook_size, codebook_dim=self.codebook_dim,
                          latent_dim=c.latent_dim, decoder_dim=self.decoder_dim, rates=self.decoder_rates,
                          max_batch=nb, max_frames=nf, device=dev,
                          encoder_dim=self.encoder_dim if keep_enc else 0, compute_dtype=compute)
            e.load_state_dict({k[len("model."):]: v for k, v in self._weights.items()})
            self._engine, self._engine_dev = e, dev
        return e

    @torch.no_grad()
    def encode(self, input_values, padding_mask=None, bandwidth=None, return_dict=None, n_quantizers=None, sample_rate=None):
        """input_values [batch, channels, samples] float → ``audio_codes`` [1, batch, n_quantizers, ceil(samples / hop)] int64
        (one chunk, like the reference with ``chunk_length=None``; modeling_dac.py:33-104). ``padding_mask`` and
        ``bandwidth`` are accepted and unused, as there; scales are ``[None]``."""
        if input_values.dim() != 3:
            raise ValueError(f"input_values must be [batch, channels, samples], got {tuple(input_values.shape)}")
        _, channels, input_length = input_values.shape
        if channels < 1 or channels > 2:
            raise ValueError(f"Number of audio channels must be 1 or 2, but got {channels}")  # modeling_dac.py:61-62
        if channels != 1:
            raise NotImplementedError("the DAC encoder is mono (Conv1d(1, ...)): pass [batch, 1, samples]")
        if sample_rate is not None and sample_rate != self.config.sampling_rate:
            raise ValueError(f"sample_rate {sample_rate} != codec sampling rate {self.config.sampling_rate}")  # dac preprocess asserts
        hop = 1
        for r in self.decoder_rates:
            hop *= int(r)
        frames = -(-input_length // hop)
        B = input_values.shape[0]
        eng = self._get_engine(B, frames, need_encoder=True)
        wave = torch.nn.functional.pad(input_values.to(self.device, torch.float32), (0, frames * hop - input_length))  # model.preprocess (:64)
        codes = torch.cat([eng.encode(wave[b0:b0 + eng.max_batch], n_quantizers) for b0 in range(0, B, eng.max_batch)], dim=0)
        encoded_frames = codes[None]
        if return_dict is False:
            return (encoded_frames, [None])
        return DACEncoderOutput(encoded_frames, [None])

    @torch.no_grad()
    def decode(self, audio_codes, audio_scales=None, padding_mask=None, return_dict=None):
        """audio_codes [1, batch, num_codebooks, frames] int64 → audio_values [batch, 1, hop*frames] float32."""
        if len(audio_codes) != 1:
            raise ValueError(f"Expected one frame, got {len(audio_codes)}")  # modeling_dac.py:135-136
        codes = audio_codes[0]
        if codes.dim() != 3:
            raise ValueError(f"audio_codes must be [1, batch, codebooks, frames], got {tuple(audio_codes.shape)}")
        B, _, T = codes.shape
        if T > 0 and (int(codes.max()) >= self.config.codebook_size or int(codes.min()) < 0):
            raise ValueError("audio_codes contain ids outside [0, codebook_size)")
        eng = self._get_engine(B, T)
        out = torch.empty(B, 1, eng.hop * T, dtype=torch.float32, device=self.device)
        for b0 in range(0, B, eng.max_batch):
            out[b0:b0 + eng.max_batch] = eng.decode(codes[b0:b0 + eng.max_batch])
        if return_dict is False:
            return (out,)
        return DACDecoderOutput(out)

    @torch.no_grad()
    def decode_filtered(self, audio_codes):
        """Not in the reference wrapper: the per-sample tail of ``generate()`` (modeling_parler_tts.py:3615-3647) in one pass. audio_codes
        [1, batch, num_codebooks, frames] possibly holding special ids (>= codebook_size): per utterance every frame with a special id is
        dropped, the rest is decoded, the waveforms are zero-padded to a common length. Returns (audio_values [batch, 1, hop*frames] float32
        - zero beyond each utterance's length -, kept-frame counts int32 [batch] on the device). The reference loops over the samples with one
        ``decode`` each; here the filter is one kernel and the codec one RAGGED pass (``ptts_dac_compact_codes`` + ``ptts_dac_decode_ragged``)."""
        if len(audio_codes) != 1:
            raise ValueError(f"Expected one frame, got {len(audio_codes)}")
        codes = audio_codes[0]
        B, _, T = codes.shape
        eng = self._get_engine(B, T)
        out = torch.empty(B, 1, eng.hop * T, dtype=torch.float32, device=self.device)
        frames = torch.empty(B, dtype=torch.int32, device=self.device)
        for b0 in range(0, B, eng.max_batch):
            cc, fr = eng.compact_codes(codes[b0:b0 + eng.max_batch])
            out[b0:b0 + eng.max_batch] = eng.decode_ragged(cc, fr)
            frames[b0:b0 + eng.max_batch] = fr
        return out, frames

    @torch.no_grad()
    def decode_chunk(self, audio_codes, first_frame: int, n_frames: Optional[int] = None, halo: int = 16, out=None, n_emit=None):
        """Streaming decode (not in the reference wrapper; its streamer re-decodes the whole cache, streamer.py:119-122):
        audio_codes [1, 1, num_codebooks, frames] → the samples of frames [first_frame, first_frame + n_frames) as
        [1, 1, hop*n_frames], computed from a window with ``halo`` frames of left context (``ptts_dac_decode_chunk``)."""
        codes = audio_codes[0]
        B, _, T = codes.shape
        n_frames = T - first_frame if n_frames is None else n_frames
        eng = self._get_engine(B, min(T, n_frames + halo), whole_batch=True)  # window = n_frames + halo frames, every utterance in one pass
        return DACDecoderOutput(eng.decode_chunk(codes, first_frame, n_frames, halo, out=out, n_emit=n_emit))

    def forward(self, tensor):
        raise ValueError("`DACModel.forward` not implemented yet")  # modeling_dac.py:144-145
