"""``ParlerTTSStreamer`` with the reference's constructor, ``put``/``end``/iterator protocol and threading model
(parler_tts/streamer.py:11-147): ``generate(..., streamer=s)`` runs in a background thread and the caller iterates
numpy chunks. The un-delay + DAC decode of the token cache every ``play_steps`` columns is kept, so the emitted
samples are the reference's; the decode itself runs on the HIP DAC engine.
"""
from __future__ import annotations

import math
from queue import Queue
from typing import Optional

import numpy as np
import torch

from .modeling_parler_tts import apply_delay_pattern_mask, build_delay_pattern_mask


def receptive_halo_frames(rates) -> int:
    """Conservative one-sided receptive field of the DAC decoder in latent frames: final k7 conv, three dilated (1, 3, 9)
    k7 residual units per block, a k=2s/stride-s transposed conv per block, the first k7 conv (13 for strides 8,8,4,2)."""
    r = 3.0
    for s in reversed(tuple(rates)):
        r += 3 * (1 + 3 + 9)
        r = math.ceil((r + 2 * s) / s)
    return int(r + 3) + 1


class ParlerTTSStreamer:
    def __init__(self, model, device: Optional[str] = None, play_steps: Optional[int] = 10, stride: Optional[int] = None,
                 timeout: Optional[float] = None, incremental: bool = True):
        self.decoder = model.decoder
        self.audio_encoder = model.audio_encoder
        self.generation_config = model.generation_config
        self.device = device if device is not None else model.device
        self.use_audio_scales = model.use_audio_scales
        self.use_4dim_audio_codes = model.use_4dim_audio_codes
        self.audio_kwargs = {"audio_scales": [None]} if self.use_audio_scales else {}
        self.play_steps = play_steps
        if stride is not None:
            self.stride = stride
        else:  # streamer.py:56-57
            hop_length = math.floor(self.audio_encoder.config.sampling_rate / self.audio_encoder.config.frame_rate)
            self.stride = hop_length * (play_steps - self.decoder.num_codebooks) // 6
        # incremental=True: every `put` decodes only the frames whose samples can still change or have not been
        # emitted, plus a receptive-field halo on the left — the emitted samples are those of the reference's full
        # re-decode (O(n^2) over an utterance) at O(n). incremental=False keeps the literal re-decode.
        self.incremental = incremental
        self.hop_length = int(math.prod(getattr(self.audio_encoder, "decoder_rates", (8, 8, 4, 2))))
        self.halo_frames = receptive_halo_frames(getattr(self.audio_encoder, "decoder_rates", (8, 8, 4, 2)))
        self.token_cache = None
        self.to_yield = 0
        self.audio_queue: Queue = Queue()
        self.stop_signal = None
        self.timeout = timeout

    def _valid_codes(self, input_ids):
        """Un-delay the raw token cache and drop frames that contain special ids (streamer.py:66-106)."""
        K = self.decoder.num_codebooks
        gc = self.generation_config
        _, mask = build_delay_pattern_mask(input_ids[:, :1], bos_token_id=gc.bos_token_id, pad_token_id=gc.decoder_start_token_id,
                                           max_length=input_ids.shape[-1], num_codebooks=K)
        input_ids = apply_delay_pattern_mask(input_ids, mask)
        keep = (mask != gc.bos_token_id) & (mask != gc.pad_token_id)
        codes = input_ids[keep].reshape(1, K, -1).to(self.audio_encoder.device)
        ok = (codes[0] >= self.audio_encoder.config.codebook_size).sum(dim=0) == 0
        return codes[:, :, ok]

    def _decode_from(self, codes, start_sample: int) -> np.ndarray:
        """Samples [start_sample:] of decode(codes): only the window that can influence them is decoded — frames from
        (start_sample // hop - halo) on; the first `halo` frames of the window absorb the left zero-padding."""
        n = codes.shape[-1]
        if n == 0:
            return np.zeros(0, dtype=np.float32)
        f0 = start_sample // self.hop_length
        if n - f0 <= 0:  # nothing new (e.g. stride 0 and no valid frame since the last put): an empty tail, as the windowed decode returns
            return np.zeros(0, dtype=np.float32)
        if hasattr(self.audio_encoder, "decode_chunk"):  # native chunk entry: reads the window in place (ptts_dac_decode_chunk)
            out = self.audio_encoder.decode_chunk(codes[None, ...], f0, n - f0, self.halo_frames).audio_values
            return out[0, 0, start_sample - f0 * self.hop_length:].cpu().float().numpy()
        w0 = max(0, f0 - self.halo_frames)
        out = self.audio_encoder.decode(audio_codes=codes[:, :, w0:][None, ...], **self.audio_kwargs).audio_values
        return out[0, 0, max(0, start_sample - w0 * self.hop_length):].cpu().float().numpy()

    def apply_delay_pattern_mask(self, input_ids):
        K = self.decoder.num_codebooks
        gc = self.generation_config
        # streamer.py:68-73 rebuilds the mask with decoder_start_token_id as the pad value (quirk kept)
        _, mask = build_delay_pattern_mask(input_ids[:, :1], bos_token_id=gc.bos_token_id, pad_token_id=gc.decoder_start_token_id,
                                           max_length=input_ids.shape[-1], num_codebooks=K)
        input_ids = apply_delay_pattern_mask(input_ids, mask)
        keep = (mask != gc.bos_token_id) & (mask != gc.pad_token_id)
        codes = input_ids[keep].reshape(1, K, -1).to(self.audio_encoder.device)
        ok = (codes[0] >= self.audio_encoder.config.codebook_size).sum(dim=0) == 0  # drop columns with special ids
        codes = codes[:, :, ok]
        if codes.shape[-1] == 0:
            return np.zeros(0, dtype=np.float32)
        out = self.audio_encoder.decode(audio_codes=codes[None, ...], **self.audio_kwargs).audio_values
        return out[0, 0].cpu().float().numpy()

    def put(self, value):
        batch_size = value.shape[0] // self.decoder.num_codebooks
        if batch_size > 1:
            raise ValueError("ParlerTTSStreamer only supports batch size 1")
        if self.token_cache is None:
            self.token_cache = value
        else:
            self.token_cache = torch.concatenate([self.token_cache, value[:, None]], dim=-1)
        if self.token_cache.shape[-1] % self.play_steps == 0:
            if not self.incremental:
                audio_values = self.apply_delay_pattern_mask(self.token_cache)
                self.on_finalized_audio(audio_values[self.to_yield: -self.stride])
                self.to_yield += len(audio_values) - self.to_yield - self.stride
            else:  # same slices as above, computed from the tail only
                codes = self._valid_codes(self.token_cache)
                total = codes.shape[-1] * self.hop_length
                # len(audio_values[to_yield:-stride]); quirk kept: with stride 0 the reference's slice is [to_yield:-0] = EMPTY while
                # to_yield still advances (streamer.py:121-122), i.e. the reference streamer drops the audio - so does this one
                n_emit = total - self.stride - self.to_yield if self.stride != 0 else 0
                tail = self._decode_from(codes, max(self.to_yield, 0)) if (total > max(self.to_yield, 0) and n_emit > 0) else np.zeros(0, dtype=np.float32)
                self.on_finalized_audio(tail[: max(n_emit, 0)] if self.to_yield >= 0 else np.zeros(0, dtype=np.float32))
                self.to_yield += total - self.to_yield - self.stride

    def end(self):
        """Flushes any remaining cache and appends the stop symbol."""
        if self.token_cache is None:
            audio_tail = np.zeros(self.to_yield)[self.to_yield:]
        elif not self.incremental:
            audio_tail = self.apply_delay_pattern_mask(self.token_cache)[self.to_yield:]
        else:
            audio_tail = self._decode_from(self._valid_codes(self.token_cache), max(self.to_yield, 0))
        self.on_finalized_audio(audio_tail, stream_end=True)

    def on_finalized_audio(self, audio: np.ndarray, stream_end: bool = False):
        self.audio_queue.put(audio, timeout=self.timeout)
        if stream_end:
            self.audio_queue.put(self.stop_signal, timeout=self.timeout)

    def __iter__(self):
        return self

    def __next__(self):
        value = self.audio_queue.get(timeout=self.timeout)
        if not isinstance(value, np.ndarray) and value == self.stop_signal:
            raise StopIteration()
        return value
