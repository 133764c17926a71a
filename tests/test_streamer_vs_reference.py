"""``parler_tts_amd.ParlerTTSStreamer`` against chunks emitted by the REFERENCE's own ``ParlerTTSStreamer``
(parler_tts/streamer.py:11-147), frozen in tests/golden/streamer_ref.npz by oracle/make_golden.py::gen_streamer (reference
class imported under the shims, oracle DAC as its codec, scripted token stream with an EOS id mid-stream). The reference
re-decodes the whole token cache at every ``play_steps``; here the O(n) halo-window path (``ptts_dac_decode_chunk``) must emit
the same chunks: same lengths, same samples."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLD
from oracle import dac_oracle as DA
from oracle import decoder_oracle as DO


def _gold():
    g = np.load(os.path.join(GOLD, "streamer_ref.npz"))
    lengths = g["lengths"].tolist()
    audio = g["audio"]
    chunks, o = [], 0
    for n in lengths:
        chunks.append(audio[o: o + n])
        o += n
    return g, chunks


def _feed(streamer, g):
    import parler_tts_amd as P

    K, L = 9, int(g["L"])
    raw = torch.from_numpy(g["raw"])
    first = P.build_delay_pattern_mask(torch.full((K, 1), 1025), 1025, 1024, L, K)[0]
    streamer.put(first)
    for j in range(1, L):
        streamer.put(raw[:, j])
    streamer.end()
    return [c for c in streamer]


def _gc():
    return types.SimpleNamespace(bos_token_id=1025, pad_token_id=1024, eos_token_id=1024, decoder_start_token_id=1025)


@pytest.mark.parametrize("incremental", [False, True])
def test_streamer_chunks_equal_reference_streamer_cpu_codec(incremental):
    from parler_tts_amd.streamer import ParlerTTSStreamer

    g, want = _gold()
    dac = DA.DacOracle(DA.DAC_TINY, DA.make_dac_weights(DA.DAC_TINY, seed=int(g["dac_seed"])))
    hop = DA.DAC_TINY.hop_length

    class Codec:
        config = types.SimpleNamespace(sampling_rate=hop * 86, frame_rate=86, codebook_size=1024, num_codebooks=9)
        device = torch.device("cpu")
        decoder_rates = DA.DAC_TINY.decoder_rates

        def decode(self, audio_codes, audio_scales=None):
            return types.SimpleNamespace(audio_values=dac.decode(audio_codes[0]))

        def decode_chunk(self, audio_codes, first_frame, n_frames=None, halo=16):  # ptts_dac_decode_chunk semantics
            codes = audio_codes[0]
            n_frames = codes.shape[-1] - first_frame if n_frames is None else n_frames
            w0 = max(0, first_frame - halo)
            wav = dac.decode(codes[:, :, w0: first_frame + n_frames])
            return types.SimpleNamespace(audio_values=wav[:, :, (first_frame - w0) * hop:])

    model = types.SimpleNamespace(decoder=types.SimpleNamespace(num_codebooks=9), audio_encoder=Codec(), generation_config=_gc(),
                                  device=torch.device("cpu"), use_audio_scales=True, use_4dim_audio_codes=True)
    st = ParlerTTSStreamer(model, play_steps=int(g["play_steps"]), incremental=incremental)
    assert st.stride == int(g["stride"])
    got = _feed(st, g)
    assert [len(c) for c in got] == [len(c) for c in want]
    for a, b in zip(got, want):
        assert np.allclose(a, b, atol=2e-6), float(np.abs(a - b).max())


@pytest.mark.gpu
@pytest.mark.parametrize("incremental", [False, True])
def test_streamer_chunks_equal_reference_streamer_hip_codec(incremental):
    import cases as C
    import parler_tts_amd as P

    g, want = _gold()
    m, *_ = C.tiny_model(seed=0)
    # the codec of the golden stream: folded-format weights of seed 4321 (tiny_model draws the parametrized format + an encoder,
    # which consumes the generator differently)
    m.audio_encoder.load_state_dict({"model." + k: v for k, v in DA.make_dac_weights(DA.DAC_TINY, seed=int(g["dac_seed"])).items()})
    m = m.to("cuda")
    m.generation_config = _gc()
    # the tiny codec's config keeps the 44.1 kHz sampling-rate default, so the default stride formula (streamer.py:56-57) would
    # not give the golden run's 58 samples: pass the stride the reference streamer used
    st = P.ParlerTTSStreamer(m, device="cuda", play_steps=int(g["play_steps"]), stride=int(g["stride"]), incremental=incremental)
    got = _feed(st, g)
    assert [len(c) for c in got] == [len(c) for c in want]
    for a, b in zip(got, want):
        assert np.allclose(a, b, atol=1e-4), float(np.abs(a - b).max())


@pytest.mark.gpu
@pytest.mark.parametrize("spec_name,T,first,n,halo,dtype", [("tiny", 50, 0, 50, 26, "f32"), ("tiny", 60, 35, 25, 26, "f32"), ("44k", 120, 60, 43, 13, "f32"),
                                                             ("44k", 120, 100, 20, 20, "f32"), ("44k", 120, 60, 43, 13, "bf16"), ("44k", 150, 0, 150, 13, "bf16")])
def test_dac_decode_chunk_equals_full_decode_window(spec_name, T, first, n, halo, dtype):
    """ptts_dac_decode_chunk: the samples of frames [first, first + n) from a window with `halo` frames of left context equal
    the same samples of a decode of frames [0, first + n) (no right context in either: the streaming situation). The halo is
    the decoder's one-sided receptive field in latent frames (streamer.receptive_halo_frames): 13 for the 44.1 kHz stack
    (strides 8, 8, 4, 2), 26 for the tiny test stack (strides 4, 2, 2, 2: smaller strides, wider field in frames)."""
    from parler_tts_amd.streamer import receptive_halo_frames

    assert receptive_halo_frames((8, 8, 4, 2)) == 13 and receptive_halo_frames((4, 2, 2, 2)) == 26
    from parler_tts_amd.engine import DacEngine
    from parler_tts_amd.synthetic import random_dac_state_dict

    if spec_name == "tiny":
        spec, dsd = DA.DAC_TINY, DA.make_dac_weights(DA.DAC_TINY, seed=4321)
        dac = DacEngine(num_codebooks=spec.num_codebooks, codebook_size=spec.codebook_size, codebook_dim=spec.codebook_dim, latent_dim=spec.latent_dim,
                        decoder_dim=spec.decoder_dim, rates=spec.decoder_rates, max_batch=2, max_frames=T)
    else:
        spec, dsd = DA.DAC_44KHZ, random_dac_state_dict(seed=4321)
        # bf16: the LDS-tiled k7 / transposed-conv kernels (128-frame tiles: T = 150 spans two tiles at the first block's rate and has
        # a ragged last tile everywhere); per-output arithmetic does not depend on the tile position, so the equality is exact there too
        dac = DacEngine(max_batch=2, max_frames=T, compute_dtype=torch.bfloat16 if dtype == "bf16" else torch.float32)
    dac.load_state_dict({k: v.cuda() for k, v in dsd.items()})
    codes = torch.randint(0, 1024, (2, 9, T), generator=torch.Generator().manual_seed(3)).cuda()
    hop = dac.hop
    full = dac.decode(codes[:, :, : first + n].contiguous())[:, :, first * hop:]
    chunk = dac.decode_chunk(codes, first, n, halo)
    assert chunk.shape == full.shape == (2, 1, n * hop)
    assert float((chunk - full).abs().max()) <= 1e-5, float((chunk - full).abs().max())
    with pytest.raises(ValueError, match="bad chunk"):
        dac.decode_chunk(codes, T - 3, 5, halo)


def test_stride_zero_end_with_nothing_new_posts_the_stop_signal():
    """stride 0 (explicit, or play_steps == num_codebooks): when `end()` finds no frame beyond what the last `put` emitted, the native
    chunk entry must not be called with an empty window (ptts_dac_decode_chunk rejects n_frames < 1 - `end()` would raise before
    posting the stop signal and the consumer would block until its timeout). The incremental and the literal re-decode paths must
    yield the same chunks, an empty final one included (with stride 0 the reference's `audio_values[to_yield:-stride]` is the empty
    slice `[to_yield:-0]` while `to_yield` advances, streamer.py:121-122: every chunk is empty - quirk kept by both paths). A second run
    with stride 1 and a stream that ends ON a boundary exercises the non-empty chunks + the empty-window guard."""
    from parler_tts_amd.streamer import ParlerTTSStreamer

    g, _ = _gold()
    dac = DA.DacOracle(DA.DAC_TINY, DA.make_dac_weights(DA.DAC_TINY, seed=int(g["dac_seed"])))
    hop = DA.DAC_TINY.hop_length

    class Codec:
        config = types.SimpleNamespace(sampling_rate=hop * 86, frame_rate=86, codebook_size=1024, num_codebooks=9)
        device = torch.device("cpu")
        decoder_rates = DA.DAC_TINY.decoder_rates

        def decode(self, audio_codes, audio_scales=None):
            return types.SimpleNamespace(audio_values=dac.decode(audio_codes[0]))

        def decode_chunk(self, audio_codes, first_frame, n_frames=None, halo=16):
            assert n_frames is None or n_frames >= 1, "ptts_dac_decode_chunk rejects an empty window"
            codes = audio_codes[0]
            n_frames = codes.shape[-1] - first_frame if n_frames is None else n_frames
            w0 = max(0, first_frame - halo)
            wav = dac.decode(codes[:, :, w0: first_frame + n_frames])
            return types.SimpleNamespace(audio_values=wav[:, :, (first_frame - w0) * hop:])

    model = types.SimpleNamespace(decoder=types.SimpleNamespace(num_codebooks=9), audio_encoder=Codec(), generation_config=_gc(),
                                  device=torch.device("cpu"), use_audio_scales=True, use_4dim_audio_codes=True)
    import parler_tts_amd as P

    K, play = 9, 12
    L = 2 * play  # the stream ends exactly on a play_steps boundary: nothing new at end()
    raw = torch.randint(0, 1024, (K, L), generator=torch.Generator().manual_seed(3))
    for stride in (0, 1):
        outs = []
        for incremental in (False, True):
            st = ParlerTTSStreamer(model, play_steps=play, stride=stride, incremental=incremental, timeout=5.0)
            st.put(P.build_delay_pattern_mask(torch.full((K, 1), 1025), 1025, 1024, L, K)[0])
            for j in range(1, L):
                st.put(raw[:, j])
            st.end()
            outs.append([c for c in st])  # raises queue.Empty after 5 s if the stop signal was never posted
        assert [len(c) for c in outs[0]] == [len(c) for c in outs[1]], stride
        assert len(outs[1]) == 3 and (stride != 0 or all(len(c) == 0 for c in outs[1]))
        for a, b in zip(*outs):
            assert np.allclose(a, b, atol=2e-6)
