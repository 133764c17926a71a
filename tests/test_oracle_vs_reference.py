"""CPU, only where the reference tree is mounted (this build container; skipped on the GPU box, which has no
/root/reference): the oracle is re-checked LIVE against the reference's own classes - the same comparisons
oracle/make_golden.py runs before it freezes tests/golden/*.npz. Nothing is written (bytecode writing is disabled by the
import shim, golden files are not touched)."""
import pytest
import torch

from oracle.reference_shims import reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="/root/reference is not mounted here")


@pytest.fixture(scope="module")
def ref():
    from oracle.reference_shims import import_reference

    return import_reference()


@pytest.mark.parametrize("variant", ["sin", "rope", "gqa"])
def test_decoder_forward_matches_reference_live(ref, variant):
    import oracle.make_golden as mg

    torch.manual_seed(0)
    worst = mg.gen_decoder(ref, variant, save=False)  # ParlerTTSForCausalLM: prefill with padded masks + 6 cached steps, SDPA and eager
    assert worst < 2e-6


def test_delay_pattern_and_eos_gate_match_reference_live(ref):
    from oracle import decoder_oracle as DO
    import parler_tts_amd as P

    M = ref.modeling_parler_tts
    for ci, (K, seq_len, max_len, bsz) in enumerate([(9, 1, 30, 2), (4, 3, 8, 1), (9, 13, 40, 1), (9, 1, 12, 1), (2, 1, 3, 2)]):
        g = torch.Generator().manual_seed(ci)
        ids = torch.randint(0, 1024, (bsz * K, seq_len), generator=g)
        ids[:, 0] = 1025
        a = M.build_delay_pattern_mask(ids, 1025, 1024, max_len, K)
        for impl in (DO.build_delay_pattern_mask, P.build_delay_pattern_mask):  # oracle restatement and the product's closed form
            b = impl(ids, 1025, 1024, max_len, K)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (ci, impl.__module__)
    K, bsz, V = 9, 2, 1088
    proc = M.ParlerTTSLogitsProcessor(1024, K, bsz, "cpu")
    gate, mine = DO.EosGate(1024, K, bsz), P.ParlerTTSLogitsProcessor(1024, K, bsz, "cpu")
    g = torch.Generator().manual_seed(3)
    seq = torch.full((bsz * K, 1), 1025, dtype=torch.long)
    for step in range(20):
        nxt = torch.randint(0, 1024, (bsz * K,), generator=g)
        nxt[torch.rand(bsz * K, generator=g) < 0.15] = 1024  # sprinkle EOS
        seq = torch.cat([seq, nxt[:, None]], dim=1)
        r = proc(seq, torch.zeros(bsz * K, V))
        assert torch.equal(r, gate(seq, torch.zeros(bsz * K, V))) and torch.equal(r, mine(seq, torch.zeros(bsz * K, V))), step


@pytest.mark.parametrize("pca,desc_mask,prompt_mask", [(False, True, True), (False, False, False), (True, True, True), (True, True, False), (True, False, True)])
def test_generate_conditioning_matches_reference_helpers_live(ref, pca, desc_mask, prompt_mask):
    """what generate() hands to the engine's prefill (description states, their mask, prompt states, their mask)
    against the reference's OWN `_prepare_text_encoder_kwargs_for_generation` (:3048-3097) and
    `_prepare_prompt_kwargs_for_generation` (:3099-3134), called unbound on a stand-in `self` that shares this package's torch
    modules (T5 encoder, enc_to_dec_proj, embed_prompts) and uses the reference's own sinusoidal positional embedding class:
    enc_to_dec_proj (T5 width != decoder width), masked description positions zeroed, prompt_cross_attention concatenation with
    positions and the synthesised all-ones masks."""
    import types

    from transformers import GenerationConfig, T5Config

    import parler_tts_amd as P

    M = ref.modeling_parler_tts
    torch.manual_seed(3)
    t5 = T5Config(vocab_size=128, d_model=96, d_kv=32, d_ff=256, num_layers=2, num_heads=4, feed_forward_proj="gated-gelu")
    dec = P.ParlerTTSDecoderConfig(vocab_size=1088, max_position_embeddings=256, num_hidden_layers=2, ffn_dim=256, num_attention_heads=2,
                                   hidden_size=128, num_codebooks=9, pad_token_id=1024, eos_token_id=1024, bos_token_id=1025)
    cfg = P.ParlerTTSConfig.from_sub_models_config(t5, P.DACConfig(latent_dim=64, decoder_dim=256, decoder_rates=[4, 2, 2, 2]), dec,
                                                   vocab_size=128, prompt_cross_attention=pca)
    m = P.ParlerTTSForConditionalGeneration(cfg).eval()
    assert hasattr(m, "enc_to_dec_proj")
    g = torch.Generator().manual_seed(4)
    B, N, Pn = 3, 7, 5
    desc, prompt_ids = torch.randint(3, 128, (B, N), generator=g), torch.randint(3, 128, (B, Pn), generator=g)
    dm = torch.ones(B, N, dtype=torch.long)
    dm[1, 5:] = 0
    dm[2, 3:] = 0
    pm = torch.ones(B, Pn, dtype=torch.long)
    pm[2, :2] = 0
    dm, pm = (dm if desc_mask else None), (pm if prompt_mask else None)

    captured = {}

    class Stop(Exception):
        pass

    class Capture:
        cfg = types.SimpleNamespace(max_batch=64, max_enc=4096, max_prompt=4096, max_ctx=1 << 20)

        def set_gen_params(self, **kw):
            pass

        def set_audio_prefix(self, codes):
            pass

        def prefill(self, enc, enc_mask, prompt, prompt_mask, sample=True):
            captured.update(enc=enc, enc_mask=enc_mask, prompt=prompt, prompt_mask=prompt_mask)
            raise Stop()

    m._get_engine = lambda *a, **k: Capture()
    with pytest.raises(Stop):
        m.generate(input_ids=desc, attention_mask=dm, prompt_input_ids=prompt_ids, prompt_attention_mask=pm, do_sample=False, max_new_tokens=4)

    stub = types.SimpleNamespace(text_encoder=m.text_encoder, get_text_encoder=lambda: m.text_encoder, decoder=types.SimpleNamespace(config=types.SimpleNamespace(hidden_size=dec.hidden_size, cross_attention_hidden_size=None)),
                                 enc_to_dec_proj=m.enc_to_dec_proj, embed_prompts=m.embed_prompts, prompt_cross_attention=pca, device=torch.device("cpu"),
                                 embed_positions=M.ParlerTTSSinusoidalPositionalEmbedding(dec.max_position_embeddings, dec.hidden_size))
    cls = M.ParlerTTSForConditionalGeneration
    with torch.no_grad():
        kw = {"attention_mask": dm, "prompt_attention_mask": pm}
        kw = cls._prepare_text_encoder_kwargs_for_generation(stub, desc, kw, "input_ids", GenerationConfig())
        kw = cls._prepare_prompt_kwargs_for_generation(stub, prompt_ids, kw)
    want_enc = kw["encoder_outputs"].last_hidden_state

    def same(a, b):
        return (a is None and b is None) or (a is not None and b is not None and a.shape == b.shape and torch.allclose(a.float(), b.float(), atol=1e-6))

    assert same(captured["enc"], want_enc)
    assert same(captured["enc_mask"], kw["attention_mask"])
    assert same(captured["prompt"], kw.get("prompt_hidden_states"))
    assert same(captured["prompt_mask"], kw.get("prompt_attention_mask"))
    if pca:
        assert captured["prompt"] is None and captured["enc"].shape[1] == N + Pn


@pytest.mark.parametrize("given", ["none", "codes", "codes_with_bos"])
def test_decoder_start_columns_match_reference_helper_live(ref, given):
    """the decoder columns generate() starts from - BOS column, plus user `decoder_input_ids` with or without a
    leading BOS column - and their delayed form (the first thing handed to a streamer, :3533-3534) against the reference's
    `_prepare_decoder_input_ids_for_generation` (:2988-3046) followed by its `build_delay_pattern_mask` (:3523-3530)."""
    import types

    from transformers import T5Config

    import parler_tts_amd as P

    M = ref.modeling_parler_tts
    torch.manual_seed(5)
    t5 = T5Config(vocab_size=128, d_model=128, d_kv=32, d_ff=256, num_layers=2, num_heads=4, feed_forward_proj="gated-gelu")
    dec = P.ParlerTTSDecoderConfig(vocab_size=1088, max_position_embeddings=256, num_hidden_layers=2, ffn_dim=256, num_attention_heads=2,
                                   hidden_size=128, num_codebooks=9, pad_token_id=1024, eos_token_id=1024, bos_token_id=1025)
    m = P.ParlerTTSForConditionalGeneration(P.ParlerTTSConfig.from_sub_models_config(
        t5, P.DACConfig(latent_dim=64, decoder_dim=256, decoder_rates=[4, 2, 2, 2]), dec, vocab_size=128)).eval()
    g = torch.Generator().manual_seed(6)
    B, K, T = 2, 9, 5
    desc, prompt_ids = torch.randint(3, 128, (B, 6), generator=g), torch.randint(3, 128, (B, 4), generator=g)
    codes = torch.randint(0, 1024, (B * K, T), generator=g)
    user = None if given == "none" else (codes if given == "codes" else torch.cat([torch.full((B * K, 1), 1025), codes], dim=1))
    max_new = 6

    class Stop(Exception):
        pass

    class Capture:
        cfg = types.SimpleNamespace(max_batch=64, max_enc=4096, max_prompt=4096, max_ctx=1 << 20)

        def set_gen_params(self, **kw):
            self.max_length = kw["max_length"]

        def set_audio_prefix(self, codes):
            self.prefix = codes

        def prefill(self, *a, **k):
            raise Stop()

    class FirstPut:
        def __init__(self):
            self.first = None

        def put(self, v):
            if self.first is None:
                self.first = v.clone()

        def end(self):
            pass

    eng, st = Capture(), FirstPut()
    m._get_engine = lambda *a, **k: eng
    with pytest.raises(Stop):
        m.generate(input_ids=desc[:1], prompt_input_ids=prompt_ids[:1], decoder_input_ids=None if user is None else user[:K],
                   do_sample=False, max_new_tokens=max_new, streamer=st)
    stub = types.SimpleNamespace(decoder=types.SimpleNamespace(num_codebooks=K), prompt_cross_attention=True, device=torch.device("cpu"),
                                 _get_decoder_start_token_id=lambda a, b: a if a is not None else b)
    kw = {} if user is None else {"decoder_input_ids": user[:K].clone()}
    ids, _ = M.ParlerTTSForConditionalGeneration._prepare_decoder_input_ids_for_generation(stub, 1, "input_ids", kw, decoder_start_token_id=1025,
                                                                                         bos_token_id=1025, device=torch.device("cpu"))
    n_given = ids.shape[1]
    assert n_given == (1 if user is None else 1 + T)
    max_length = max_new + n_given  # :3458-3469: max_new_tokens counts from the given decoder columns
    want, _ = ref.build_delay_pattern_mask(ids, 1025, 1024, max_length, K)
    assert eng.max_length == max_length
    assert st.first is not None and torch.equal(st.first, want)
    if user is None:
        assert eng.prefix is None
    else:
        assert torch.equal(eng.prefix.cpu(), codes[:K])


def _product_model_with_oracle_engine(pca: bool):
    """tests/test_generate_glue_cpu.py::_model with a `prompt_cross_attention` switch: this package's model, oracle-backed engine."""
    import types

    from transformers import T5Config

    import parler_tts_amd as P
    import test_generate_glue_cpu as G
    from oracle import dac_oracle as DA
    from oracle import decoder_oracle as DO

    torch.manual_seed(0)
    t5 = T5Config(vocab_size=128, d_model=128, d_kv=32, d_ff=256, num_layers=2, num_heads=4, feed_forward_proj="gated-gelu")
    dec = P.ParlerTTSDecoderConfig(vocab_size=1088, max_position_embeddings=256, num_hidden_layers=2, ffn_dim=256, num_attention_heads=2,
                                   hidden_size=128, num_codebooks=9, pad_token_id=1024, eos_token_id=1024, bos_token_id=1025)
    m = P.ParlerTTSForConditionalGeneration(P.ParlerTTSConfig.from_sub_models_config(
        t5, P.DACConfig(latent_dim=64, decoder_dim=256, decoder_rates=[4, 2, 2, 2]), dec, vocab_size=128, prompt_cross_attention=pca)).eval()
    spec, sd = DO.TINY, DO.make_decoder_weights(DO.TINY, seed=1237)
    for k in range(9):
        sd[f"lm_heads.{k}.weight"][1024] *= 6.0  # EOS reachable with random heads
    m.decoder.load_state_dict(sd, strict=False)
    eng = G.OracleEngine(spec, sd)
    m._get_engine = lambda B, N, Pp, L, T=0: eng
    dac = DA.DacOracle(DA.DAC_TINY, DA.make_dac_weights(DA.DAC_TINY, seed=4321))
    m.audio_encoder.decode = lambda audio_codes, audio_scales=None, **kw: types.SimpleNamespace(audio_values=dac.decode(audio_codes[0].cpu()))
    m.audio_encoder.decode_filtered = None  # CPU stand-in codec: generate() then runs the reference's per-sample loop literally (:3627-3647)
    return m, spec, sd, eng


@pytest.mark.parametrize("masks,pca,prefix", [(False, False, False), (True, False, False), (True, True, False), (False, True, False), (True, False, True)])
def test_generate_ids_equal_a_loop_of_the_reference_generate_time_functions_live(ref, masks, pca, prefix):
    """End to end without the reference's `generate()` (written against transformers 4.46's GenerationMixin, not runnable on the
    installed 5.x): every model-specific step of it is the reference's OWN function, called unbound on a stand-in `self` -
    `_prepare_text_encoder_kwargs_for_generation`, `_prepare_prompt_kwargs_for_generation`, `_prepare_decoder_input_ids_for_generation`,
    `build_delay_pattern_mask`, `_get_initial_cache_position`, and per step `prepare_inputs_for_generation` (:2882-2986: delay mask,
    last-column slicing, prompt dropped after step 0, decoder_attention_mask synthesised from the prompt mask, cache positions) ->
    `ParlerTTSForConditionalGeneration.forward` (:2695-2880) -> the reference ParlerTTSForCausalLM with an EncoderDecoderCache ->
    MinNewTokens + the reference `ParlerTTSLogitsProcessor`; only the third-party `_sample` skeleton (pinned separately against the
    installed transformers) is restated. The ids must equal the ones `generate()` of this package produces on the same inputs
    (host glue + oracle-backed engine; the HIP engine is pinned to the oracle by the -m gpu tests). Variants: padded description +
    left-padded prompt masks, `prompt_cross_attention`, and a voice-prompt prefix given as `decoder_input_ids` (multi-column first pass,
    lengths counted from the given columns)."""
    import math
    import types

    from transformers import GenerationConfig
    from transformers.cache_utils import DynamicCache, EncoderDecoderCache

    import oracle.make_golden as mg

    M = ref.modeling_parler_tts
    C = M.ParlerTTSForConditionalGeneration
    m, spec, sd, eng = _product_model_with_oracle_engine(pca)
    K, bos, pad, eos = spec.num_codebooks, spec.bos_token_id, spec.pad_token_id, spec.eos_token_id
    g = torch.Generator().manual_seed(21 + 2 * int(pca) + int(prefix))
    B, N, Pn, T0, min_new = 2, 8, 5, (4 if prefix else 0), 3
    L = 34 + T0
    desc, prompt_ids = torch.randint(3, 128, (B, N), generator=g), torch.randint(3, 128, (B, Pn), generator=g)
    codes = torch.randint(0, 1024, (B * K, T0), generator=g) if prefix else None
    dm = pm = None
    if masks:
        dm = torch.ones(B, N, dtype=torch.long); dm[1, 5:] = 0
        pm = torch.ones(B, Pn, dtype=torch.long); pm[1, :2] = 0  # left-padded prompt, like a batched tokenizer call
    # --- this package ---------------------------------------------------------------------------------------------------------------
    m.generate(input_ids=desc, attention_mask=dm, prompt_input_ids=prompt_ids, prompt_attention_mask=pm, decoder_input_ids=codes, do_sample=False,
               max_length=L, min_new_tokens=min_new)
    mine = eng.full  # [B*K, columns]: the delayed ids of the whole run
    # --- the reference's functions -------------------------------------------------------------------------------------------------
    lm = mg.build_reference_lm(ref, spec, sd)
    lm.config.cross_attention_hidden_size = None
    stub = types.SimpleNamespace(
        text_encoder=m.text_encoder, get_text_encoder=lambda: m.text_encoder, decoder=lm, embed_prompts=m.embed_prompts, prompt_cross_attention=pca,
        embed_positions=M.ParlerTTSSinusoidalPositionalEmbedding(256, spec.hidden_size),
        device=torch.device("cpu"), config=types.SimpleNamespace(use_return_dict=True, decoder=types.SimpleNamespace(audio_channels=1)),
        generation_config=types.SimpleNamespace(bos_token_id=bos, pad_token_id=pad, max_length=L),
        _get_decoder_start_token_id=lambda a, b: a if a is not None else b)
    with torch.no_grad():
        kw = {"attention_mask": dm, "prompt_attention_mask": pm}
        if prefix:
            kw["decoder_input_ids"] = codes.clone()
        kw = C._prepare_text_encoder_kwargs_for_generation(stub, desc, kw, "input_ids", GenerationConfig())
        kw = C._prepare_prompt_kwargs_for_generation(stub, prompt_ids, kw)
        ids, kw = C._prepare_decoder_input_ids_for_generation(stub, B, "input_ids", kw, decoder_start_token_id=bos, bos_token_id=bos,
                                                              device=torch.device("cpu"))
        given = ids.shape[1]
        assert given == 1 + T0
        seq, pattern = lm.build_delay_pattern_mask(ids, bos_token_id=bos, pad_token_id=pad, max_length=L)
        kw["decoder_delay_pattern_mask"] = pattern
        kw["past_key_values"] = EncoderDecoderCache(DynamicCache(), DynamicCache())
        kw["use_cache"] = True
        kw = C._get_initial_cache_position(stub, seq, kw)
        assert kw["cache_position"].tolist() == list(range((0 if pca else Pn) + given))  # prompt positions (unless cross-attended) + given columns
        proc = M.ParlerTTSLogitsProcessor(eos, K, B, "cpu")
        unfinished = torch.ones(B * K, dtype=torch.long)
        while True:
            inputs = C.prepare_inputs_for_generation(stub, seq, **kw)
            out = C.forward(stub, **inputs, return_dict=True)
            scores = out.logits[:, -1, :].clone().float()
            if seq.shape[-1] - given < min_new:
                scores[:, eos] = -math.inf
            scores = proc(seq, scores)
            nxt = torch.argmax(scores, dim=-1)
            nxt = nxt * unfinished + pad * (1 - unfinished)
            seq = torch.cat([seq, nxt[:, None]], dim=-1)
            kw["past_key_values"] = out.past_key_values
            kw["cache_position"] = kw["cache_position"][-1:] + 1  # GenerationMixin._update_model_kwargs_for_generation
            unfinished = unfinished & ~((nxt == eos) | (seq.shape[-1] >= L)).long()
            if unfinished.max() == 0:
                break
    assert seq.shape == mine.shape and seq.shape[1] > given + 8, (seq.shape, mine.shape)
    assert torch.equal(seq, mine)
    assert int((seq[:, given:] == eos).any(dim=1).sum()) >= 2  # EOS really happened on some rows: the gate and the padding were exercised


def test_delay_pattern_closed_form_equals_reference_function_property_live(ref):
    """hypothesis: this package's closed-form build_delay_pattern_mask against the reference's loop-built one (modeling:214-276) over
    random codebook counts, batch sizes, given columns and max lengths (the domain the reference's loop accepts, :241-243)."""
    hypothesis = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st

    import parler_tts_amd as P

    @settings(max_examples=120, deadline=None)
    @given(K=st.integers(1, 9), bsz=st.integers(1, 3), seq_len=st.integers(1, 12), extra=st.integers(0, 30), seed=st.integers(0, 10_000))
    def check(K, bsz, seq_len, extra, seed):
        max_len = seq_len + extra
        hypothesis.assume(max_len < 2 * K - 1 or max_len >= seq_len + K - 1)
        ids = torch.randint(0, 1024, (bsz * K, seq_len), generator=torch.Generator().manual_seed(seed))
        ids[:, 0] = 1025
        a = ref.build_delay_pattern_mask(ids, 1025, 1024, max_len, K)
        b = P.build_delay_pattern_mask(ids, 1025, 1024, max_len, K)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])

    check()


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_streamer_equals_reference_streamer_on_random_streams_live(ref, seed):
    """The golden stream (tests/golden/streamer_ref.npz) generalised: random stream length, `play_steps`, explicit or default stride
    and 0-2 special ids mid-stream; the REFERENCE's own ParlerTTSStreamer (full re-decode of the cache every `play_steps`,
    streamer.py:66-131) and this package's incremental halo-window streamer, both over the oracle codec, must emit the same chunks."""
    import types

    import numpy as np
    from parler_tts.streamer import ParlerTTSStreamer as RefStreamer

    import oracle.make_golden as mg
    from oracle import dac_oracle as DA
    from oracle import decoder_oracle as DO
    from parler_tts_amd.streamer import ParlerTTSStreamer

    g = torch.Generator().manual_seed(100 + seed)
    spec = DO.TINY
    K = spec.num_codebooks
    L = int(torch.randint(30, 70, (1,), generator=g))
    play_steps = int(torch.randint(10, 24, (1,), generator=g))
    stride = None if seed % 2 == 0 else int(torch.randint(0, 40, (1,), generator=g))
    dec = mg.build_reference_lm(ref, spec, DO.make_decoder_weights(spec, seed=1234))
    dac = DA.DacOracle(DA.DAC_TINY, DA.make_dac_weights(DA.DAC_TINY, seed=4321))
    hop = DA.DAC_TINY.hop_length

    class Codec:
        config = types.SimpleNamespace(sampling_rate=hop * 86, frame_rate=86, codebook_size=1024, num_codebooks=K)
        device = torch.device("cpu")
        decoder_rates = DA.DAC_TINY.decoder_rates

        def decode(self, audio_codes, audio_scales=None):
            return types.SimpleNamespace(audio_values=dac.decode(audio_codes[0]))

        def decode_chunk(self, audio_codes, first_frame, n_frames=None, halo=16):  # ptts_dac_decode_chunk semantics
            codes = audio_codes[0]
            n_frames = codes.shape[-1] - first_frame if n_frames is None else n_frames
            w0 = max(0, first_frame - halo)
            return types.SimpleNamespace(audio_values=dac.decode(codes[:, :, w0: first_frame + n_frames])[:, :, (first_frame - w0) * hop:])

    gc = types.SimpleNamespace(bos_token_id=spec.bos_token_id, pad_token_id=spec.pad_token_id, eos_token_id=spec.eos_token_id,
                               decoder_start_token_id=spec.bos_token_id)
    raw = torch.randint(0, 1024, (K, L), generator=g)
    for _ in range(seed % 3):  # special ids mid-stream: the frame that holds one is dropped (streamer.py:95-104)
        raw[int(torch.randint(0, K, (1,), generator=g)), int(torch.randint(12, L - 1, (1,), generator=g))] = spec.eos_token_id
    first, _ = dec.build_delay_pattern_mask(torch.full((K, 1), spec.bos_token_id), bos_token_id=spec.bos_token_id, pad_token_id=spec.pad_token_id,
                                            max_length=L)
    outs = []
    for cls, decoder in ((RefStreamer, dec), (ParlerTTSStreamer, types.SimpleNamespace(num_codebooks=K))):
        model = types.SimpleNamespace(decoder=decoder, audio_encoder=Codec(), generation_config=gc, device=torch.device("cpu"),
                                      use_audio_scales=True, use_4dim_audio_codes=True)
        st = cls(model, play_steps=play_steps, stride=stride)
        st.put(first.clone())
        for j in range(1, L):
            st.put(raw[:, j])
        st.end()
        outs.append([np.asarray(c) for c in st])
    want, got = outs
    assert len(want) >= 2 and [len(c) for c in got] == [len(c) for c in want], ([len(c) for c in got], [len(c) for c in want])
    for a, b in zip(got, want):  # window decode vs whole-cache decode on the CPU conv kernels: fp32 summation order differs at the 1e-6 level
        assert np.allclose(a, b, atol=2e-5), float(np.abs(a - b).max())


def test_bf16_oracle_sits_between_reference_fp32_and_reference_bf16_live(ref):
    """the timed dtype (bf16) pinned against the reference's OWN bf16 arithmetic. The reference's documented inference mode is
    `model.to(device, dtype=torch.bfloat16)` (INFERENCE.md:24-32): bf16 weights, bf16 residual stream / LayerNorm outputs / probabilities
    (modeling_parler_tts.py:983-1074). The engine (and DecoderOracle(precision="bf16"), which rounds where the engine rounds) keeps the residual
    stream, LayerNorm and softmax in fp32 and rounds only the GEMM operands and the K/V cache. Mini-v1 widths at FULL depth (24 layers), prefill of 9
    positions + 8 cached steps, the reference's ParlerTTSForCausalLM run three ways on the same weights:
        ref_fp32 (the truth), ref_bf16 (the reference's own bf16 run), oracle_bf16 (the engine's rounding model)
    asserted: the engine's rounding model is at least as close to the fp32 truth as the reference's own bf16 run is, and it stays within the
    reference-bf16 noise of the reference's bf16 run (they are two roundings of the same function). The three numbers are printed (pytest -s) and
    recorded in profiles/r06_parity_bf16_reference.txt."""
    import copy

    from transformers.cache_utils import DynamicCache, EncoderDecoderCache

    import oracle.make_golden as mg
    from oracle import decoder_oracle as DO

    spec = DO.DecoderSpec(num_hidden_layers=24, max_position_embeddings=256)  # Mini-v1 widths: H = 1024, 16 heads, F = 4096
    sd = DO.make_decoder_weights(spec, seed=2024)
    bsz, N, P, steps = 1, 21, 8, 8
    enc, enc_mask, prompt, prompt_mask = mg.synth_inputs(spec, bsz, N, P, seed=5, padded=False)
    g = torch.Generator().manual_seed(17)
    K = spec.num_codebooks
    ids0 = torch.full((bsz * K, 1), spec.bos_token_id, dtype=torch.long)
    step_ids = torch.randint(0, 1024, (steps, bsz * K, 1), generator=g)

    def run_reference(dtype):
        m = mg.build_reference_lm(ref, spec, sd)
        if dtype != torch.float32:
            m = m.to(dtype)
        e, p = enc.to(dtype), prompt.to(dtype)
        cache = EncoderDecoderCache(DynamicCache(), DynamicCache())
        out = [mg.reference_forward(m, cache, ids0, e, enc_mask, p, prompt_mask, 0)[:, -1].float()]
        past = P + 1
        for s in range(steps):
            out.append(mg.reference_forward(m, cache, step_ids[s], e, enc_mask, p, prompt_mask, past)[:, -1].float())
            past += 1
        del m
        return torch.stack(out)

    ref32 = run_reference(torch.float32)
    ref16 = run_reference(torch.bfloat16)
    orc = DO.DecoderOracle(spec, sd, precision="bf16")
    o = [orc.forward(ids0, enc, enc_mask, prompt, prompt_mask)[:, -1]]
    for s in range(steps):
        o.append(orc.forward(step_ids[s])[:, -1])
    o16 = torch.stack(o)
    d_ref = float((ref16 - ref32).abs().max())      # the reference's own bf16 run vs its fp32 run
    d_orc = float((o16 - ref32).abs().max())        # the engine's rounding model vs the reference in fp32
    d_cross = float((o16 - ref16).abs().max())      # the engine's rounding model vs the reference in bf16
    rms = lambda t: float(t.pow(2).mean().sqrt())
    flips_ref = int((ref16.argmax(-1) != ref32.argmax(-1)).sum())
    flips_orc = int((o16.argmax(-1) != ref32.argmax(-1)).sum())
    print(f"\n[bf16 pin, Mini-v1 widths x 24 layers, prefill {P + 1} positions + {steps} cached steps, {ref32.shape[0] * ref32.shape[1]} logit rows] "
          f"max|ref_bf16 - ref_fp32| = {d_ref:.3e} (rms {rms(ref16 - ref32):.3e}, argmax flips {flips_ref})  "
          f"max|oracle_bf16 - ref_fp32| = {d_orc:.3e} (rms {rms(o16 - ref32):.3e}, argmax flips {flips_orc})  "
          f"max|oracle_bf16 - ref_bf16| = {d_cross:.3e} (rms {rms(o16 - ref16):.3e})")
    assert d_orc <= d_ref, (d_orc, d_ref)
    assert rms(o16 - ref32) <= rms(ref16 - ref32)
    assert d_cross <= 2.0 * d_ref, (d_cross, d_ref)
