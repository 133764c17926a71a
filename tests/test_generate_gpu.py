"""GPU end-to-end: ParlerTTSForConditionalGeneration.generate() (T5 encoder in stock PyTorch-ROCm → HIP decoder engine
→ un-delay → HIP DAC engine) against the oracle pipeline on the same weights and inputs.
Bars: greedy ids bit-exact (via identical waveform support), waveform RMS <= 1e-4 (fp32)."""
import threading

import numpy as np
import pytest
import torch

from oracle import dac_oracle as DA
from oracle import decoder_oracle as DO

pytestmark = pytest.mark.gpu


def _tiny_model(seed=0, eos_gain=None, rope=False, prompt_cross_attention=False):
    import parler_tts_amd as P
    from transformers import T5Config

    torch.manual_seed(seed)
    t5 = T5Config(vocab_size=128, d_model=128, d_kv=32, d_ff=256, num_layers=2, num_heads=4, feed_forward_proj="gated-gelu")
    dec = P.ParlerTTSDecoderConfig(vocab_size=1088, max_position_embeddings=256, num_hidden_layers=2, ffn_dim=256, num_attention_heads=2,
                                   hidden_size=128, num_codebooks=9, pad_token_id=1024, eos_token_id=1024, bos_token_id=1025, rope_embeddings=rope)
    dac = P.DACConfig(latent_dim=64, decoder_dim=256, decoder_rates=[4, 2, 2, 2], encoder_dim=16)
    cfg = P.ParlerTTSConfig.from_sub_models_config(t5, dac, dec, vocab_size=128, prompt_cross_attention=prompt_cross_attention)
    m = P.ParlerTTSForConditionalGeneration(cfg)
    spec = DO.DecoderSpec(**{**DO.TINY.__dict__, "rope_embeddings": rope})
    sd = DO.make_decoder_weights(spec, seed=1234 + seed)
    if eos_gain:
        for k in range(9):
            sd[f"lm_heads.{k}.weight"][1024] *= eos_gain
    else:  # fixed-length runs: a trained model never emits the 64 padding ids >= codebook_size; random heads would, and
        for k in range(9):  # generate() (like the reference :3627-3636) drops every frame that contains one
            sd[f"lm_heads.{k}.weight"][1024:] = 0.0
    m.decoder.load_state_dict(sd, strict=False)
    dsd = DA.make_dac_weights(DA.DAC_TINY, seed=4321, weight_norm_format="parametrized", with_encoder=True)
    m.audio_encoder.load_state_dict({"model." + k: v for k, v in dsd.items()})
    return m, spec, sd, dsd


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
    # seed 3: margin-safe on this path (min top-2 margin 2.6e-4), 4 rows reach EOS, the two samples keep 6 and 7 frames
    m, spec, sd, dsd = _tiny_model(seed=3, eos_gain=6.0)
    m = m.to("cuda")
    g = torch.Generator().manual_seed(1)
    desc = torch.randint(3, 128, (2, 9), generator=g)
    desc_mask = torch.ones(2, 9, dtype=torch.long)
    desc_mask[1, 6:] = 0
    prompt_ids = torch.randint(3, 128, (2, 5), generator=g)
    prompt_mask = torch.ones(2, 5, dtype=torch.long)
    prompt_mask[1, :2] = 0
    gp = DO.GenParams(max_length=41, min_new_tokens=10)
    tr, wav_ref = _oracle_pipeline(m, spec, sd, dsd, desc, desc_mask, prompt_ids, prompt_mask, gp)
    if tr.min_margin < 1e-4:
        pytest.skip(f"seed not margin-safe ({tr.min_margin:.1e})")
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


@pytest.mark.parametrize("rope,pca", [(False, False), (True, False), (False, True)])
def test_generate_fixed_length_variants(rope, pca):
    """min_new_tokens == max_new_tokens (the benchmark's deterministic-length setting), RoPE and prompt_cross_attention."""
    m, spec, sd, dsd = _tiny_model(seed=2, rope=rope, prompt_cross_attention=pca)
    m = m.to("cuda")
    g = torch.Generator().manual_seed(3)
    desc = torch.randint(3, 128, (1, 7), generator=g)
    prompt_ids = torch.randint(3, 128, (1, 4), generator=g)
    wav = m.generate(input_ids=desc.cuda(), prompt_input_ids=prompt_ids.cuda(), do_sample=False, max_new_tokens=30, min_new_tokens=30)
    F_ = 30 + 1 - 9
    assert wav.shape == (1, F_ * DA.DAC_TINY.hop_length)
    if not pca:
        gp = DO.GenParams(max_length=31, min_new_tokens=30)
        tr, wav_ref = _oracle_pipeline(m, spec, sd, dsd, desc, None, prompt_ids, None, gp)
        if tr.min_margin >= 1e-4:
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
    streamer = P.ParlerTTSStreamer(m, device="cuda", play_steps=20, stride=8)
    th = threading.Thread(target=m.generate, kwargs=dict(streamer=streamer, **kw))
    th.start()
    chunks = [c for c in streamer]
    th.join()
    audio = np.concatenate(chunks)
    assert len(chunks) >= 3 and audio.shape == full.shape
    # the final flush decodes the complete token cache: its tail must equal the non-streamed waveform's tail
    n = len(chunks[-1])
    assert n > 0 and np.allclose(audio[-n:], full[-n:], atol=1e-5)
    # incremental (halo-window) decoding emits exactly the chunks of the reference's full re-decode of the whole cache
    chunks_by_mode = []
    for inc in (True, False):
        st = P.ParlerTTSStreamer(m, device="cuda", play_steps=20, stride=8, incremental=inc)
        th = threading.Thread(target=m.generate, kwargs=dict(streamer=st, **kw))
        th.start()
        chunks_by_mode.append([c for c in st])
        th.join()
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
    """`input_values` (voice prompt, modeling:3136-3194): DAC-encode on the HIP engine, continue the codes, decode everything.
    Checked against the oracle pipeline fed the SAME prefix codes (ids margin-safe -> identical frames, waveform RMS <= 1e-4),
    and `decoder_input_ids` (with or without a leading BOS column) must give the same audio as `input_values`."""
    m, spec, sd, dsd = _tiny_model(seed=0)
    m = m.to("cuda")
    g = torch.Generator().manual_seed(4)
    desc = torch.randint(3, 128, (1, 8), generator=g)
    prompt_ids = torch.randint(3, 128, (1, 5), generator=g)
    voice = 0.3 * torch.randn(1, 1, 32 * 6 - 5, generator=g)  # 6 frames after the preprocess padding
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
    # oracle pipeline on the same prefix
    enc = m._encode_description(desc.cuda(), None).float().cpu()
    prompt = m.embed_prompts(prompt_ids.cuda()).float().cpu()
    gp = DO.GenParams(max_length=27, min_new_tokens=20)
    tr = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, None, prompt, None, gp, decoder_input_ids=ids2.cpu())
    if tr.min_margin > 2e-4:
        ref_codes = DO.undelay(tr.sequences, spec, 27, decoder_input_ids=ids2.cpu())
        assert torch.equal(ref_codes[0, :, :6], ids2.cpu())  # the voice prompt survives the delay / un-delay round trip intact
        wav_ref = DA.DacOracle(DA.DAC_TINY, dsd).decode(DO.valid_frames(ref_codes[0])[None])[0, 0]
        assert wav_ref.shape[0] == out.sequences.shape[1]
        assert float((out.sequences[0].cpu() - wav_ref).pow(2).mean().sqrt()) <= 1e-4
    with pytest.raises(ValueError, match="no room"):
        m.generate(decoder_input_ids=ids2, input_ids=desc.cuda(), prompt_input_ids=prompt_ids.cuda(), max_length=7)


def test_bf16_model_runs_and_tracks_fp32():
    m, *_ = _tiny_model(seed=2)
    desc = torch.randint(3, 128, (1, 7), generator=torch.Generator().manual_seed(3))
    prompt_ids = torch.randint(3, 128, (1, 4), generator=torch.Generator().manual_seed(4))
    kw = dict(do_sample=False, max_new_tokens=20, min_new_tokens=20)
    m = m.to("cuda")
    a = m.generate(input_ids=desc.cuda(), prompt_input_ids=prompt_ids.cuda(), **kw)
    m = m.to(dtype=torch.bfloat16)
    b = m.generate(input_ids=desc.cuda(), prompt_input_ids=prompt_ids.cuda(), **kw)
    assert a.shape == b.shape and b.dtype == torch.float32 and torch.isfinite(b).all()
