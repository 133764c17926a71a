"""GPU end-to-end: ParlerTTSForConditionalGeneration.generate() (T5 encoder in stock PyTorch-ROCm → HIP decoder engine
→ un-delay → HIP DAC engine) against the oracle pipeline on the same weights and inputs.
Bars: greedy ids bit-exact (via identical waveform support), waveform RMS <= 1e-4 (fp32)."""
import threading

import numpy as np
import pytest
import torch

import cases as C
from oracle import dac_oracle as DA
from oracle import decoder_oracle as DO

pytestmark = pytest.mark.gpu


_tiny_model = C.tiny_model


def _oracle_pipeline(m, spec, sd, dsd, desc, desc_mask, prompt_ids, prompt_mask, gp):
    """Same conditioning tensors as the product path (computed by the same torch T5 on the GPU), then everything
    downstream from the oracle on CPU."""
    enc = m._encode_description(desc.cuda(), desc_mask.cuda() if desc_mask is not None else None).float().cpu()
    prompt = m.embed_prompts(prompt_ids.cuda()).float().cpu()
    tr = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, desc_mask, prompt, prompt_mask, gp)
    codes = DO.undelay(tr.sequences, spec, gp.max_length)
    wavs = []
    for b in range(codes.shape[0]):
        c = DO.valid_frames(codes[b])
        wavs.append(DA.DacOracle(DA.DAC_TINY, dsd).decode(c[None])[0, 0] if c.shape[1] else torch.zeros(1))
    return tr, wavs


def test_generate_greedy_matches_oracle_pipeline_with_eos_and_padding():
    ms, isd = C.GEN_EOS_SEEDS  # scanned on the oracle: margin-safe, 4 rows reach EOS, the two samples keep 6 and 5 frames
    m, spec, sd, dsd = _tiny_model(seed=ms, eos_gain=6.0)
    m = m.to("cuda")
    desc, desc_mask, prompt_ids, prompt_mask, gp = C.gen_eos_inputs(isd)
    tr, wav_ref = _oracle_pipeline(m, spec, sd, dsd, desc, desc_mask, prompt_ids, prompt_mask, gp)
    assert tr.min_margin >= C.MARGIN
    assert int((tr.sequences == 1024).any(1).sum()) >= 2  # the EOS / padding paths are really exercised
    out = m.generate(input_ids=desc.cuda(), attention_mask=desc_mask.cuda(), prompt_input_ids=prompt_ids.cuda(),
                     prompt_attention_mask=prompt_mask.cuda(), do_sample=False, max_new_tokens=40, min_new_tokens=10,
                     return_dict_in_generate=True)
    wav, lens = out.sequences.cpu(), out["audios_length"]
    assert wav.dim() == 2 and wav.shape[0] == 2
    for b in range(2):
        assert lens[b] == wav_ref[b].shape[0], (b, lens, wav_ref[b].shape)
        err = (wav[b, : lens[b]] - wav_ref[b]).pow(2).mean().sqrt()
        assert float(err) <= 1e-4, float(err)
        assert float(wav[b, lens[b]:].abs().sum()) == 0.0  # zero padding beyond each sample's length


def _oracle_pipeline_pca(m, spec, sd, dsd, desc, prompt_ids, gp):
    """prompt_cross_attention (:3102-3128): the prompt embeddings (+ sinusoidal positions) are appended to the description
    states and nothing is prepended to the decoder input: restated on the oracle side from the same torch modules."""
    enc = m._encode_description(desc.cuda(), None).float().cpu()
    ph = m.embed_prompts(prompt_ids.cuda()).float().cpu()
    ph = ph + m.embed_positions.weights[: ph.shape[1]].float().cpu()[None]
    enc = torch.cat([enc, ph], dim=1)
    tr = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, None, None, None, gp)
    codes = DO.undelay(tr.sequences, spec, gp.max_length)
    c = DO.valid_frames(codes[0])
    return tr, [DA.DacOracle(DA.DAC_TINY, dsd).decode(c[None])[0, 0]]


@pytest.mark.parametrize("rope,pca", [(False, False), (True, False), (False, True)])
def test_generate_fixed_length_variants(rope, pca):
    """min_new_tokens == max_new_tokens (the benchmark's deterministic-length setting), RoPE and prompt_cross_attention:
    waveform vs the oracle pipeline in all three variants (seeds scanned on the oracle)."""
    ms, isd = C.GEN_FIXED_SEEDS[rope]
    m, spec, sd, dsd = _tiny_model(seed=ms, rope=rope, prompt_cross_attention=pca)
    m = m.to("cuda")
    desc, prompt_ids, gp = C.gen_fixed_inputs(isd)
    wav = m.generate(input_ids=desc.cuda(), prompt_input_ids=prompt_ids.cuda(), do_sample=False, max_new_tokens=30, min_new_tokens=30)
    F_ = 30 + 1 - 9
    assert wav.shape == (1, F_ * DA.DAC_TINY.hop_length)
    if pca:
        tr, wav_ref = _oracle_pipeline_pca(m, spec, sd, dsd, desc, prompt_ids, gp)
    else:
        tr, wav_ref = _oracle_pipeline(m, spec, sd, dsd, desc, None, prompt_ids, None, gp)
    assert tr.min_margin >= C.MARGIN, tr.min_margin  # scanned: 5.5e-4 / 3.5e-4 (rope) / 1.9e-4 (prompt_cross_attention)
    assert float((wav[0].cpu() - wav_ref[0]).pow(2).mean().sqrt()) <= 1e-4
    assert torch.isfinite(wav).all() and float(wav.abs().max()) <= 1.0


def test_custom_logits_processor_list_equals_default_path():
    import parler_tts_amd as P

    m, *_ = _tiny_model(seed=6, eos_gain=6.0)
    m = m.to("cuda")
    g = torch.Generator().manual_seed(5)
    desc = torch.randint(3, 128, (2, 6), generator=g).cuda()
    prompt_ids = torch.randint(3, 128, (2, 3), generator=g).cuda()
    kw = dict(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_new_tokens=36, min_new_tokens=4)
    a = m.generate(**kw)
    procs = [P.ParlerTTSLogitsProcessor(1024, 9, 2, "cuda")]  # what generate() installs by default (reference :3418)
    b = m.generate(logits_processor=procs, **kw)
    assert a.shape == b.shape and torch.equal(a, b)

    class StopAt20:
        def __call__(self, input_ids, scores):
            return torch.full((input_ids.shape[0],), input_ids.shape[-1] >= 20, dtype=torch.bool, device=input_ids.device)

    c = m.generate(stopping_criteria=[StopAt20()], logits_processor=procs[:0] or [P.ParlerTTSLogitsProcessor(1024, 9, 2, "cuda")], **kw)
    assert c.shape[1] <= (20 - 9) * DA.DAC_TINY.hop_length


def test_stopping_criteria_only_keeps_the_default_eos_gate():
    """generate(stopping_criteria=[...]) with NO logits_processor: the reference still installs the default
    LogitsProcessorList([ParlerTTSLogitsProcessor]) (:3418). With a criterion that never fires the output must equal the
    default device-loop path bit for bit (EOS-heavy model: without the gate codebook k would emit EOS before k-1)."""
    ms, isd = C.GEN_EOS_SEEDS
    m, spec, sd, dsd = _tiny_model(seed=ms, eos_gain=6.0)
    m = m.to("cuda")
    desc, desc_mask, prompt_ids, prompt_mask, gp = C.gen_eos_inputs(isd)
    kw = dict(input_ids=desc.cuda(), attention_mask=desc_mask.cuda(), prompt_input_ids=prompt_ids.cuda(),
              prompt_attention_mask=prompt_mask.cuda(), do_sample=False, max_new_tokens=40, min_new_tokens=10)

    class Never:
        def __call__(self, input_ids, scores):
            return torch.zeros(input_ids.shape[0], dtype=torch.bool, device=input_ids.device)

    a = m.generate(**kw)
    b = m.generate(stopping_criteria=[Never()], **kw)
    assert a.shape == b.shape and torch.equal(a, b)
    tr, wav_ref = _oracle_pipeline(m, spec, sd, dsd, desc, desc_mask, prompt_ids, prompt_mask, gp)
    for i in range(2):
        n = wav_ref[i].shape[0]
        assert float((b[i, :n].cpu() - wav_ref[i]).pow(2).mean().sqrt()) <= 1e-4


def test_generation_config_processors_run_the_host_loop_on_the_hip_engine():
    """GenerationConfig options the device sampler does not implement (repetition_penalty, no_repeat_ngram_size ...) switch generate() to
    the host loop with transformers' own processor objects (generation_extras.py; the reference gets them from `_get_logits_processor`,
    :3540-3547). Same HIP engine, same logits: the call must equal, bit for bit, the same processors handed in as a user list (a path the
    tests above cover), and differ from the plain call. What the processors compute is pinned on CPU (tests/test_generation_extras_cpu.py)."""
    import parler_tts_amd as P
    from transformers.generation.logits_process import NoRepeatNGramLogitsProcessor, RepetitionPenaltyLogitsProcessor

    m, *_ = _tiny_model(seed=6, eos_gain=6.0)
    m = m.to("cuda")
    g = torch.Generator().manual_seed(5)
    desc = torch.randint(3, 128, (2, 6), generator=g).cuda()
    prompt_ids = torch.randint(3, 128, (2, 3), generator=g).cuda()
    kw = dict(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_new_tokens=36, min_new_tokens=4)
    plain = m.generate(**kw)
    a = m.generate(repetition_penalty=1.4, no_repeat_ngram_size=2, **kw)
    user = [RepetitionPenaltyLogitsProcessor(penalty=1.4), NoRepeatNGramLogitsProcessor(2), P.ParlerTTSLogitsProcessor(1024, 9, 2, "cuda")]
    b = m.generate(logits_processor=user, **kw)
    assert a.shape == b.shape and torch.equal(a, b)
    assert a.shape != plain.shape or not torch.equal(a, plain)
    with pytest.raises(NotImplementedError, match="guidance_scale"):
        m.generate(guidance_scale=3.0, **kw)


def test_sampling_is_seeded_and_in_range():
    m, *_ = _tiny_model(seed=4)
    m = m.to("cuda")
    desc = torch.randint(3, 128, (2, 6)).cuda()
    prompt_ids = torch.randint(3, 128, (2, 3)).cuda()
    kw = dict(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=True, temperature=0.9, top_k=40, max_new_tokens=24, min_new_tokens=24)
    torch.manual_seed(7)
    a = m.generate(**kw)
    torch.manual_seed(7)
    b = m.generate(**kw)
    torch.manual_seed(8)
    c = m.generate(**kw)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert a.shape == (2, (24 + 1 - 9) * DA.DAC_TINY.hop_length)


def test_streamer_thread_protocol():
    import parler_tts_amd as P

    m, *_ = _tiny_model(seed=2)
    m = m.to("cuda")
    g = torch.Generator().manual_seed(3)
    desc = torch.randint(3, 128, (1, 7), generator=g).cuda()
    prompt_ids = torch.randint(3, 128, (1, 4), generator=g).cuda()
    kw = dict(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_new_tokens=120, min_new_tokens=120)
    full = m.generate(**kw)[0].cpu().numpy()
    def run(st):  # a generate() thread that dies must not leave the consumer blocked on the queue: record the error, end the stream
        errs = []

        def target():
            try:
                m.generate(streamer=st, **kw)
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
                st.on_finalized_audio(np.zeros(0, dtype=np.float32), stream_end=True)

        th = threading.Thread(target=target)
        th.start()
        out = [c for c in st]
        th.join(timeout=120)
        assert not th.is_alive(), "generate() thread still running"
        if errs:
            raise errs[0]
        return out

    streamer = P.ParlerTTSStreamer(m, device="cuda", play_steps=20, stride=8, timeout=120)
    chunks = run(streamer)
    audio = np.concatenate(chunks)
    assert len(chunks) >= 3 and audio.shape == full.shape
    # the final flush decodes the complete token cache: its tail must equal the non-streamed waveform's tail
    n = len(chunks[-1])
    assert n > 0 and np.allclose(audio[-n:], full[-n:], atol=1e-5)
    # incremental (halo-window) decoding emits exactly the chunks of the reference's full re-decode of the whole cache
    chunks_by_mode = []
    for inc in (True, False):
        st = P.ParlerTTSStreamer(m, device="cuda", play_steps=20, stride=8, incremental=inc, timeout=120)
        chunks_by_mode.append(run(st))
    assert [len(c) for c in chunks_by_mode[0]] == [len(c) for c in chunks_by_mode[1]]
    for a_, b_ in zip(*chunks_by_mode):
        assert np.allclose(a_, b_, atol=1e-6)
    with pytest.raises(ValueError, match="batch size 1"):
        m.generate(input_ids=desc.repeat(2, 1), prompt_input_ids=prompt_ids.repeat(2, 1), streamer=P.ParlerTTSStreamer(m, play_steps=20), max_new_tokens=12)


def test_num_return_sequences_expands_batch():
    m, *_ = _tiny_model(seed=2)
    m = m.to("cuda")
    desc = torch.randint(3, 128, (2, 6), generator=torch.Generator().manual_seed(1)).cuda()
    prompt_ids = torch.randint(3, 128, (2, 3), generator=torch.Generator().manual_seed(2)).cuda()
    # GenerationConfig rejects greedy + num_return_sequences > 1: sample with top_k=1 (deterministic, equals greedy)
    kw = dict(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=True, top_k=1, max_new_tokens=20, min_new_tokens=20)
    a = m.generate(**kw)
    b = m.generate(num_return_sequences=2, **kw)
    assert b.shape[0] == 4 and torch.equal(b[0], b[1]) and torch.equal(b[0], a[0]) and torch.equal(b[2], a[1])


def test_voice_prompt_input_values_and_decoder_input_ids():
    """`input_values` (voice prompt, modeling:3136-3194): DAC-encode on the HIP engine, continue the codes, decode everything;
    `decoder_input_ids` (with or without a leading BOS column) must give the same audio as `input_values`. The oracle
    comparison uses a synthetic prompt given as `decoder_input_ids` (deterministic, margin-scanned on the CPU): identical
    frames, the prompt survives the delay / un-delay round trip, waveform RMS <= 1e-4."""
    ms, isd = C.GEN_VOICE_SEEDS
    m, spec, sd, dsd = _tiny_model(seed=ms)
    m = m.to("cuda")
    desc, prompt_ids, voice, syn, gp = C.gen_voice_inputs(isd)
    codes = m.audio_encoder.encode(voice.cuda()).audio_codes[0]  # [1, 9, 6]
    assert codes.shape == (1, 9, 6)
    kw = dict(input_ids=desc.cuda(), prompt_input_ids=prompt_ids.cuda(), do_sample=False, max_new_tokens=20, min_new_tokens=20)
    out = m.generate(input_values=voice.cuda(), return_dict_in_generate=True, **kw)
    # 1 BOS + 6 prompt + 20 new columns = 27 -> Lout - K = 18 frames
    assert out.sequences.shape == (1, 32 * (27 - 9)) and out["audios_length"] == [32 * 18], out.sequences.shape
    ids2 = codes.reshape(9, 6)
    w2 = m.generate(decoder_input_ids=ids2, **kw)
    w3 = m.generate(decoder_input_ids=torch.cat([torch.full((9, 1), 1025, device="cuda"), ids2], 1), **kw)
    assert torch.equal(out.sequences, w2) and torch.equal(w2, w3)
    # oracle pipeline on the synthetic prefix
    ws = m.generate(decoder_input_ids=syn.cuda(), **kw)
    enc = m._encode_description(desc.cuda(), None).float().cpu()
    prompt = m.embed_prompts(prompt_ids.cuda()).float().cpu()
    tr = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, None, prompt, None, gp, decoder_input_ids=syn)
    assert tr.min_margin >= C.MARGIN
    ref_codes = DO.undelay(tr.sequences, spec, 27, decoder_input_ids=syn)
    assert torch.equal(ref_codes[0, :, :6], syn)  # the voice prompt survives the delay / un-delay round trip intact
    wav_ref = DA.DacOracle(DA.DAC_TINY, dsd).decode(DO.valid_frames(ref_codes[0])[None])[0, 0]
    assert wav_ref.shape[0] == ws.shape[1]
    assert float((ws[0].cpu() - wav_ref).pow(2).mean().sqrt()) <= 1e-4
    with pytest.raises(ValueError, match="no room"):
        m.generate(decoder_input_ids=ids2, input_ids=desc.cuda(), prompt_input_ids=prompt_ids.cuda(), max_length=7)


def test_bf16_model_tracks_the_quantised_oracle():
    """model.to(bfloat16): decoder engine AND codec switch to their bf16 modes. (1) every token the bf16 engine chose is
    (near-)optimal under the bf16-quantised oracle's own logits at the SAME history (teacher-forced on the engine's ids, so an
    early near-tie flip cannot hide later errors): oracle logit of the chosen token >= oracle max - 3e-2; >= 90 % of the
    choices are the oracle's exact arg-max. (2) the waveform equals the fp32 DAC oracle on the engine's own codes to the
    bf16-codec tolerance (3 % of the signal RMS)."""
    ms, isd = C.GEN_FIXED_SEEDS[False]
    m, spec, sd, dsd = _tiny_model(seed=ms)
    desc, prompt_ids, gp = C.gen_fixed_inputs(isd)
    m = m.to("cuda").to(dtype=torch.bfloat16)
    b = m.generate(input_ids=desc.cuda(), prompt_input_ids=prompt_ids.cuda(), do_sample=False, max_new_tokens=30, min_new_tokens=30)
    assert b.dtype == torch.float32 and torch.isfinite(b).all() and b.shape == (1, (31 - 9) * DA.DAC_TINY.hop_length)
    ids = m._engine.ids().cpu()
    assert ids.shape == (9, 31)
    enc = m._encode_description(desc.cuda(), None).float().cpu()
    prompt = m.embed_prompts(prompt_ids.cuda()).float().cpu()
    outs = DO.teacher_forced_logits(DO.DecoderOracle(spec, sd, precision="bf16"), enc, None, prompt, None, ids, gp.max_length)
    exact, total = 0, 0
    for s_, lg in enumerate(outs):
        lg = lg.clone()
        lg[:, spec.eos_token_id] = -float("inf")  # min_new_tokens == max_new_tokens
        chosen = ids[:, s_ + 1]
        gap = lg.max(-1)[0] - lg.gather(1, chosen[:, None])[:, 0]
        assert float(gap.max()) <= 3e-2, (s_, float(gap.max()))
        exact += int((gap == 0).sum()); total += gap.numel()
    assert exact >= 0.9 * total, (exact, total)
    codes = DO.undelay(ids, spec, gp.max_length)
    wav_ref = DA.DacOracle(DA.DAC_TINY, dsd).decode(DO.valid_frames(codes[0])[None])[0, 0]
    assert wav_ref.shape[0] == b.shape[1]
    assert float((b[0].cpu() - wav_ref).pow(2).mean().sqrt()) <= 0.03 * float(wav_ref.pow(2).mean().sqrt())


def test_codec_overlapped_with_the_token_loop_gives_the_full_decode():
    """generate() without a streamer decodes finished frames chunk by chunk on a side stream while the token graph keeps
    running (ptts_dac_decode_chunk with left + right halos): the waveform must be bit-identical to the one-shot decode after
    the loop, and an EOS-terminated run (special ids -> per-sample column filter) must fall back to the sequential branch."""
    ms, isd = C.GEN_FIXED_SEEDS[False]
    m, spec, sd, dsd = _tiny_model(seed=ms)
    m = m.to("cuda")
    desc, prompt_ids, _ = C.gen_fixed_inputs(isd)
    kw = dict(input_ids=desc.cuda().repeat(2, 1), prompt_input_ids=prompt_ids.cuda().repeat(2, 1), do_sample=False, max_new_tokens=200,
              min_new_tokens=200)
    m.overlap_codec = True
    a = m.generate(**kw)
    m.overlap_codec = False
    b = m.generate(**kw)
    assert a.shape == b.shape == (2, (201 - 9) * DA.DAC_TINY.hop_length)
    assert torch.equal(a, b)
    # EOS-heavy model: ids contain special tokens -> the overlapped loop must hand over to the filtered decode
    m2, *_ = _tiny_model(seed=C.GEN_EOS_SEEDS[0], eos_gain=6.0)
    m2 = m2.to("cuda")
    d2, dm2, p2, pm2, _ = C.gen_eos_inputs(C.GEN_EOS_SEEDS[1])
    kw2 = dict(input_ids=d2.cuda(), attention_mask=dm2.cuda(), prompt_input_ids=p2.cuda(), prompt_attention_mask=pm2.cuda(), do_sample=False,
               max_new_tokens=150, min_new_tokens=10, return_dict_in_generate=True)
    m2.overlap_codec = True
    o1 = m2.generate(**kw2)
    m2.overlap_codec = False
    o2 = m2.generate(**kw2)
    assert o1["audios_length"] == o2["audios_length"] and torch.equal(o1.sequences, o2.sequences)


def test_generate_at_gemv_widths_with_enc_to_dec_proj_matches_oracle():
    """512-wide decoder behind a 128-wide text encoder: `enc_to_dec_proj` (:2388-2392) on the conditioning path, the single-
    utterance GEMV step with the folded cross-attention block and the hipGraph replay inside generate(); fp32 waveform vs the
    oracle pipeline (seed scanned: margin 3.2e-3)."""
    ms, isd = C.GEN_MID_SEEDS
    m, spec, sd, dsd = C.mid_model(ms)
    assert hasattr(m, "enc_to_dec_proj")
    m = m.to("cuda")
    desc, prompt_ids, gp = C.gen_fixed_inputs(isd)
    wav = m.generate(input_ids=desc.cuda(), prompt_input_ids=prompt_ids.cuda(), do_sample=False, max_new_tokens=30, min_new_tokens=30)
    tr, wav_ref = _oracle_pipeline(m, spec, sd, dsd, desc, None, prompt_ids, None, gp)
    assert tr.min_margin >= C.MARGIN
    assert torch.equal(m._engine.ids().cpu(), tr.sequences)
    assert float((wav[0].cpu() - wav_ref[0]).pow(2).mean().sqrt()) <= 1e-4


def test_generate_with_fp8_weights_tracks_the_quantised_oracle():
    """model.enable_fp8_weights() (BASELINE configs[4]): bf16 model, decode step streaming e4m3 weights. Every token the engine
    chose must be (near-)optimal under the oracle that evaluates the SAME quantised model at the same history; switching the
    mode off again restores the plain bf16 engine."""
    from oracle import fp8_oracle as FO

    ms, isd = C.GEN_MID_SEEDS
    m, spec, sd, dsd = C.mid_model(ms)
    desc, prompt_ids, gp = C.gen_fixed_inputs(isd)
    m = m.to("cuda").to(dtype=torch.bfloat16).enable_fp8_weights()
    b = m.generate(input_ids=desc.cuda(), prompt_input_ids=prompt_ids.cuda(), do_sample=False, max_new_tokens=30, min_new_tokens=30)
    assert m._engine.weights_fp8 and torch.isfinite(b).all() and b.shape == (1, (31 - 9) * DA.DAC_TINY.hop_length)
    ids = m._engine.ids().cpu()
    enc = m._encode_description(desc.cuda(), None).float().cpu()
    prompt = m.embed_prompts(prompt_ids.cuda()).float().cpu()
    qsd = FO.quantize_decoder_weights(sd)
    outs = DO.teacher_forced_logits(DO.DecoderOracle(spec, qsd, precision="bf16"), enc, None, prompt, None, ids, gp.max_length)
    exact, total = 0, 0
    for s_, lg in enumerate(outs):
        lg = lg.clone()
        lg[:, spec.eos_token_id] = -float("inf")
        gap = lg.max(-1)[0] - lg.gather(1, ids[:, s_ + 1][:, None])[:, 0]
        assert float(gap.max()) <= 3e-2, (s_, float(gap.max()))
        exact += int((gap == 0).sum()); total += gap.numel()
    assert exact >= 0.9 * total, (exact, total)
    m.enable_fp8_weights(False)
    m.generate(input_ids=desc.cuda(), prompt_input_ids=prompt_ids.cuda(), do_sample=False, max_new_tokens=12, min_new_tokens=12)
    assert not m._engine.weights_fp8


def test_generate_sharded_over_rccl_world_size_1_equals_generate():
    """`parler_tts_amd.generate_sharded` on the RCCL ("nccl") backend: device tensors through all_gather / gather, the weights through
    `broadcast_model_weights`. One GPU is all the build environment reaches, so the process group has ONE rank (the world-size-2
    protocol is proven over gloo in tests/test_distributed_cpu.py): the result must equal the plain `generate()` call - waveforms,
    lengths, input order - with an EOS-bearing model (ragged lengths, zero padding)."""
    import os
    import socket

    import torch.distributed as dist

    import parler_tts_amd as P

    m, *_ = C.tiny_model(seed=0, eos_gain=6.0)
    m = m.to("cuda")
    g = torch.Generator().manual_seed(1)
    desc, prompt_ids = torch.randint(3, 128, (3, 9), generator=g).cuda(), torch.randint(3, 128, (3, 4), generator=g).cuda()
    kw = dict(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_length=40, min_new_tokens=3)
    before = m.generate(return_dict_in_generate=True, **kw)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n = P.broadcast_model_weights(m, src=0)
        assert n >= 1 and m._engine is None
        # the reference point is the plain call on the SAME replica state: the broadcast moves the codec's weight-norm tensors to the
        # device, where the fold w = g * v / ||v|| rounds differently from the host-side fold of a freshly loaded model (4.6e-6 on the
        # waveform; the token ids are identical)
        want = m.generate(return_dict_in_generate=True, **kw)
        assert list(want["audios_length"]) == list(before["audios_length"]) and float((want.sequences - before.sequences).abs().max()) < 1e-4
        for dst in (0, None):
            got = P.generate_sharded(m, dst=dst, **kw)
            assert list(got["audios_length"]) == list(want["audios_length"])
            assert got.sequences.shape == want.sequences.shape and torch.equal(got.sequences, want.sequences.float().cpu())
    finally:
        dist.destroy_process_group()


def test_generate_eos_terminated_batch_of_8_goes_through_one_ragged_codec_pass():
    """generate()'s per-sample tail (modeling_parler_tts.py:3615-3647) on an EOS-terminated batch of 8 with DISTINCT lengths: the ids the token
    loop produced (captured at the hand-over) are un-delayed, filtered and decoded utterance by utterance by the ORACLE, exactly as the
    reference's loop does; generate() itself runs ONE filter kernel + ONE ragged codec pass (DACModel.decode_filtered). Waveforms <= 1e-4 RMS,
    lengths equal, zero padding exact. Independent of the LM's top-2 margins (whatever ids the engine chose are the ones both sides decode)."""
    m, spec, sd, dsd = _tiny_model(seed=2, eos_gain=6.0)
    m = m.to("cuda")
    g = torch.Generator().manual_seed(200)
    B = 8
    desc = torch.randint(3, 128, (B, 9), generator=g)
    dm = torch.ones(B, 9, dtype=torch.long)
    dm[1, 6:] = 0
    dm[5, 4:] = 0
    pid = torch.randint(3, 128, (B, 5), generator=g)
    pm = torch.ones(B, 5, dtype=torch.long)
    pm[1, :2] = 0
    seen, calls = {}, []
    tail = m._undelay_and_decode
    filt = m.audio_encoder.decode_filtered

    def spy_tail(output_ids, *a, **k):
        seen["ids"] = output_ids.detach().cpu().clone()
        return tail(output_ids, *a, **k)

    def spy_filtered(codes):
        calls.append(tuple(codes.shape))
        return filt(codes)

    m._undelay_and_decode = spy_tail
    m.audio_encoder.decode_filtered = spy_filtered
    out = m.generate(input_ids=desc.cuda(), attention_mask=dm.cuda(), prompt_input_ids=pid.cuda(), prompt_attention_mask=pm.cuda(), do_sample=False,
                     max_length=60, min_new_tokens=4, return_dict_in_generate=True)
    wav, lens = out.sequences.cpu(), out["audios_length"]
    assert len(calls) == 1 and calls[0][1] == B  # one ragged pass over the whole batch
    codes = DO.undelay(seen["ids"], spec, 60)
    orc = DA.DacOracle(DA.DAC_TINY, dsd)
    want = []
    for b in range(B):
        c = DO.valid_frames(codes[b])
        want.append(orc.decode(c[None])[0, 0] if c.shape[1] else torch.zeros(1))
    assert len({w.shape[0] for w in want}) >= 4, [w.shape[0] for w in want]  # really ragged
    assert wav.shape == (B, max(w.shape[0] for w in want))
    for b in range(B):
        assert lens[b] == want[b].shape[0], (b, lens, want[b].shape)
        assert float((wav[b, : lens[b]] - want[b]).pow(2).mean().sqrt()) <= 1e-4
        assert float(wav[b, lens[b]:].abs().sum()) == 0.0
