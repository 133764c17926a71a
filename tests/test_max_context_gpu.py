"""Maximum sizes of the released checkpoints on the GPU path: generation to `generation_config.max_length` = 30 s x 86 frames/s = 2580
columns (helpers/model_init_scripts/init_model_600M.py:29,62) behind a 33-position prompt, i.e. a self-attention context of 2612
positions - every context bucket the engine pre-captures, every KV-split count of the attention nodes, the last rows of the KV arena -
and the codec on the 2571 frames such a run returns.

The oracle side stays cheap: the engine's OWN free run (prefill + 2579 hipGraph replays) is judged column by column against ONE batched
causal forward of the oracle teacher-forced on those ids (same arithmetic as the cached passes up to summation order), so an early
near-tie cannot cascade: a choice may differ from the oracle's arg-max only inside the stated error band. The logits of the LAST pass
(context = the maximum) are compared directly. Mini-v1 widths, 2 layers (depth does not change the indexing; the 24-layer pins at
context ~460 are tests/test_bench_config_parity_gpu.py).

Tolerances: fp32 |dlogit| <= 5e-5 (summation order), bf16 <= 2e-2 against DecoderOracle(precision="bf16") (activation re-rounding, as in
tests/test_lm_gpu.py); codec exact-f32 waveform |d| <= 1e-4 absolute on the compared windows (north_star: RMS 1e-4)."""
import os

import pytest
import torch

import cases as C
from helpers import log_parity, make_dac, make_engine
from oracle import dac_oracle as DA
from oracle import decoder_oracle as DO

pytestmark = pytest.mark.gpu

MAX_LENGTH = 2580  # int(30 * frame_rate), init_model_600M.py:62
N_DESC, N_PROMPT = 64, 33
LOG = "r06_parity_max_context.txt"


def _run(dtype, prec, bsz, tol, masks, seed, kv_fp8=False, spec=None, tag=""):
    spec = spec or DO.DecoderSpec(num_hidden_layers=2)  # Mini-v1 widths, max_position_embeddings 4096
    K, L = spec.num_codebooks, MAX_LENGTH
    sd = DO.make_decoder_weights(spec, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    enc = torch.randn(bsz, N_DESC, spec.hidden_size, generator=g)
    prompt = torch.randn(bsz, N_PROMPT, spec.hidden_size, generator=g) * 0.5
    enc_mask = prompt_mask = None
    if masks:
        enc_mask, prompt_mask = C.ragged_masks(bsz, N_DESC, N_PROMPT, enc_step=2)
        enc = enc * enc_mask[..., None]
    eng = make_engine(spec, sd, dtype, max_batch=bsz, max_ctx=N_PROMPT + L, max_enc=N_DESC, max_prompt=N_PROMPT + 1, kv_fp8=kv_fp8)
    eng.set_gen_params(max_length=L, min_new_tokens=L - 1)
    ids = eng.generate_ids(enc, enc_mask, prompt, prompt_mask).cpu()
    last = eng.logits().cpu()  # the logits of pass L - 2: the one that chose the last column, at the largest context of the run
    eng.close()
    assert ids.shape == (bsz * K, L)
    assert bool((ids[:, 0] == spec.bos_token_id).all())
    _, pattern = DO.build_delay_pattern_mask(ids[:, :1], spec.bos_token_id, spec.pad_token_id, L, K)
    fed = DO.apply_delay_pattern_mask(ids, pattern)[:, : L - 1]
    torch.set_num_threads(min(os.cpu_count() or 8, 16))
    orc = DO.DecoderOracle(spec, sd, precision=prec)
    orc.kv_fp8 = kv_fp8
    lgs = []
    with torch.no_grad():
        for b in range(bsz):  # one utterance at a time: the [heads, 2612, 2612] score tensors of a whole batch would not fit a small host
            orc.reset()
            sl = slice(b, b + 1)
            lgs.append(orc.forward(fed[b * K:(b + 1) * K], enc[sl], None if enc_mask is None else enc_mask[sl], prompt[sl],
                                   None if prompt_mask is None else prompt_mask[sl])[:, -(L - 1):].float())
    lg = torch.cat(lgs, dim=0)  # [rows, pass, V]: pass s chooses column s + 1
    err_last = float((last - lg[:, -1]).abs().max())
    m = lg.clone()
    m[..., spec.eos_token_id] = -float("inf")  # min_new_tokens blocks EOS on every pass of this run
    top2, idx = torch.topk(m, 2, dim=-1)
    margin = top2[..., 0] - top2[..., 1]
    diff = idx[..., 0] != ids[:, 1:]
    # a differing choice is legitimate only if the engine's pick scores within the error band of the oracle's best
    gap = top2[..., 0] - m.gather(-1, ids[:, 1:, None])[..., 0]
    outside = int((diff & (gap > 2 * tol)).sum())
    late = diff[:, L // 2:]
    log_parity(f"[max context {prec} bs={bsz}{' e4m3-kv' if kv_fp8 else ''}{tag}] {L - 1} free-running passes x {bsz * K} rows, context up to {N_PROMPT + L - 2}: "
               f"{int(diff.sum())} of {diff.numel()} choices differ from the oracle's arg-max at the same history ({int(late.sum())} in the second half), "
               f"{outside} outside 2 x {tol:g}; max |dlogit| of the last pass {err_last:.2e}; min oracle top-2 margin {float(margin.min()):.2e}", LOG)
    assert err_last <= tol, err_last
    assert outside == 0, outside
    assert float(diff.float().mean()) <= 0.02
    return ids


def test_fp32_single_utterance_to_max_length():
    """GEMV step: 2579 graph replays, contexts 33..2611 (every bucket, every split count)."""
    _run(torch.float32, "fp32", 1, 5e-5, masks=False, seed=101)


def test_bf16_single_utterance_to_max_length():
    _run(torch.bfloat16, "bf16", 1, 2e-2, masks=False, seed=102)


def test_bf16_ragged_batch_of_12_to_max_length():
    """MFMA-strip step (> 8 utterances) with ragged description / prompt masks."""
    _run(torch.bfloat16, "bf16", 12, 2e-2, masks=True, seed=103)


def test_bf16_e4m3_kv_cache_batch_of_12_to_max_length():
    """Opt-in e4m3 KV cache (own oracle leg): the scale arena and the byte rows at the far end of the context."""
    _run(torch.bfloat16, "bf16", 12, 2e-2, masks=True, seed=104, kv_fp8=True)


@pytest.mark.parametrize("bsz", [1, 9])
def test_large_v1_widths_to_max_length(bsz):
    """Large-v1 widths (H = 1536, 24 heads, F = 6144; helpers/model_init_scripts/init_large_model.py:25-43), 2 layers, bf16: the single-utterance GEMV step without
    the folded cross block (that width runs LN2 + cross-q + attention and the output projection as two nodes) and 9 ragged utterances on the MFMA strips."""
    spec = DO.DecoderSpec(hidden_size=1536, num_hidden_layers=2, num_attention_heads=24, ffn_dim=6144)
    _run(torch.bfloat16, "bf16", bsz, 2e-2, masks=bsz > 1, seed=105 + bsz, spec=spec, tag=" Large-v1 widths")


@pytest.mark.parametrize("bsz", [1, 9])
def test_rope_and_grouped_query_attention_to_max_length(bsz):
    """`rope_embeddings` (positions up to 2611 through the cos / sin tables, the q-only rotation quirk of the cross block) with grouped-query attention
    (4 self / 2 cross K/V heads) - the configuration axes of the released Mini-v1 / Large-v1 checkpoints' siblings - fp32."""
    spec = DO.DecoderSpec(num_hidden_layers=2, rope_embeddings=True, num_key_value_heads=4, num_cross_attention_key_value_heads=2)
    _run(torch.float32, "fp32", bsz, 5e-5, masks=bsz > 1, seed=120 + bsz, spec=spec, tag=" RoPE + GQA")


def test_codec_on_the_frames_of_a_max_length_run():
    """44 kHz decoder, exact-f32 engine, 2571 frames (= 2580 - 9 columns after the un-delay) x 2 utterances in one call: three windows of the
    waveform (first, middle, last frames) against the oracle run on 64-frame windows of the same codes - away from a window's inner edges
    (one-sided receptive field 13 latent frames) the decoder sees identical inputs, and the utterance's own edges are the window's edges."""
    from parler_tts_amd.streamer import receptive_halo_frames

    spec = DA.DAC_44KHZ
    sd = DA.make_dac_weights(spec, seed=77)
    T = MAX_LENGTH - spec.num_codebooks
    codes = torch.randint(0, 1024, (2, 9, T), generator=torch.Generator().manual_seed(9))
    wav = make_dac(spec, sd, max_batch=2, max_frames=T).decode(codes.cuda()).cpu()
    hop = spec.hop_length
    assert wav.shape == (2, 1, hop * T)
    halo = receptive_halo_frames(spec.decoder_rates)
    orc = DA.DacOracle(spec, sd)
    worst = 0.0
    for a, b in ((0, 64), (T // 2 - 32, T // 2 + 32), (T - 64, T)):
        ref = orc.decode(codes[:, :, a:b])
        lo = a if a == 0 else a + halo
        hi = b if b == T else b - halo
        d = float((wav[..., lo * hop: hi * hop] - ref[..., (lo - a) * hop: (hi - a) * hop]).abs().max())
        worst = max(worst, d)
    log_parity(f"[codec, 2 x {T} frames, exact-f32] three 64-frame windows vs the oracle: max |d| {worst:.2e}", LOG)
    assert worst <= 1e-4, worst


@pytest.mark.parametrize("dtype,prec,tol", [(torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)])
def test_five_second_voice_prompts_prefilled_in_one_pass(dtype, prec, tol):
    """decoder_input_ids of 430 frames (5 s of a reference voice, modeling:3136-3194) for 9 utterances with ragged masks: the prompt, the BOS column and the
    430 delayed code columns run as ONE prefill pass of 9 x 464 rows - the > 256-row GEMMs with the K / V rows written by the QKV epilogue, the f32-MFMA prefill
    attention with eight 64-query blocks x up to eight key blocks per (utterance, head) - and the logits of the continuation's first position are compared
    with the oracle's single multi-column forward (what the reference runs)."""
    spec = DO.DecoderSpec(num_hidden_layers=2)
    K, T, bsz = spec.num_codebooks, 430, 9
    sd = DO.make_decoder_weights(spec, seed=211)
    g = torch.Generator().manual_seed(212)
    enc = torch.randn(bsz, N_DESC, spec.hidden_size, generator=g)
    prompt = torch.randn(bsz, N_PROMPT, spec.hidden_size, generator=g) * 0.5
    enc_mask, prompt_mask = C.ragged_masks(bsz, N_DESC, N_PROMPT, enc_step=2)
    enc = enc * enc_mask[..., None]
    pre = torch.randint(0, 1024, (bsz, K, T), generator=g)
    L = T + 40
    eng = make_engine(spec, sd, dtype, max_batch=bsz, max_ctx=N_PROMPT + L + 8, max_enc=N_DESC, max_prompt=N_PROMPT + 1 + T)
    eng.set_gen_params(max_length=L)
    eng.set_audio_prefix(pre)
    eng.prefill(enc, enc_mask, prompt, prompt_mask, sample=False)
    got = eng.logits().cpu()
    eng.close()
    seq0 = torch.cat([torch.full((bsz * K, 1), spec.bos_token_id), pre.reshape(bsz * K, T)], 1)
    delayed, pattern = DO.build_delay_pattern_mask(seq0, spec.bos_token_id, spec.pad_token_id, L, K)
    fed = DO.apply_delay_pattern_mask(delayed, pattern)
    torch.set_num_threads(min(os.cpu_count() or 8, 16))
    with torch.no_grad():
        ref = DO.DecoderOracle(spec, sd, precision=prec).forward(fed, enc, enc_mask, prompt, prompt_mask)[:, -1].float()
    err = float((got - ref).abs().max())
    log_parity(f"[voice prompt of {T} frames x {bsz} utterances, {prec}] one prefill pass of {bsz} x {N_PROMPT + 1 + T} rows: max |dlogit| at the continuation's first position {err:.2e}", LOG)
    assert err <= tol, err


def test_generate_end_to_end_at_the_default_max_length():
    """`ParlerTTSForConditionalGeneration.generate()` itself at the released checkpoints' `generation_config.max_length` = 2580 (tiny T5 + tiny decoder + tiny
    codec, 2 ragged utterances, EOS allowed from column 2000 on): the chunked decode loop without host synchronisation while MinNewTokens blocks EOS, every context
    bucket, the un-delay of a 2580-column run, the per-sample special-id filter and ONE ragged codec pass over ~2570 frames per utterance. A free run of 2 x 9 x 2579
    choices cannot be margin-safe, so the engine's own ids (read back from the engine generate() used) are judged against the oracle's batched forward at the same
    history, and everything behind the ids - un-delay, filter, codec, zero padding, `audios_length` - against the oracle pipeline on those ids."""
    m, spec, sd, dsd = C.tiny_model(seed=3, eos_gain=6.0, max_positions=4096)
    m = m.to("cuda")
    L, K = MAX_LENGTH, spec.num_codebooks
    desc, desc_mask, prompt_ids, prompt_mask, _ = C.gen_eos_inputs(111)
    m.generation_config.max_length = L
    out = m.generate(input_ids=desc.cuda(), attention_mask=desc_mask.cuda(), prompt_input_ids=prompt_ids.cuda(), prompt_attention_mask=prompt_mask.cuda(),
                     do_sample=False, min_new_tokens=2000, return_dict_in_generate=True)
    wav, lens = out.sequences.cpu(), [int(x) for x in out["audios_length"]]
    ids = m._engine.ids().cpu()
    Lout = ids.shape[1]
    assert ids.shape[0] == 2 * K and 2001 <= Lout <= L
    # (a) the ids against the oracle at the same history
    enc = m._encode_description(desc.cuda(), desc_mask.cuda()).float().cpu()
    prompt = m.embed_prompts(prompt_ids.cuda()).float().cpu()
    _, pattern = DO.build_delay_pattern_mask(ids[:, :1], spec.bos_token_id, spec.pad_token_id, L, K)
    fed = DO.apply_delay_pattern_mask(ids, pattern)[:, : Lout - 1]
    with torch.no_grad():
        lg = DO.DecoderOracle(spec, sd).forward(fed, enc, desc_mask, prompt, prompt_mask)[:, -(Lout - 1):].float()
    # judged on the 1999 passes in which MinNewTokens blocks EOS (every row alive, plain arg-max); the EOS gate, finished-row padding and the stop test
    # behind them are pinned bit-exactly by the short margin-safe runs of tests/test_generate_gpu.py and tests/test_lm_gpu.py
    n_judged = 2000 - 1
    m_ = lg[:, :n_judged].clone()
    m_[..., spec.eos_token_id] = -float("inf")
    top2, idx = torch.topk(m_, 2, dim=-1)
    chosen = ids[:, 1: 1 + n_judged]
    gap = top2[..., 0] - m_.gather(-1, chosen[..., None])[..., 0]
    bad = int(((idx[..., 0] != chosen) & (gap > 1e-4)).sum())
    # (b) everything behind the ids against the oracle pipeline on those ids
    codes = DO.undelay(ids, spec, L)
    worst = 0.0
    for b in range(2):
        c = DO.valid_frames(codes[b])
        ref = DA.DacOracle(DA.DAC_TINY, dsd).decode(c[None])[0, 0]
        assert lens[b] == ref.shape[0], (b, lens, ref.shape)
        worst = max(worst, float((wav[b, : lens[b]] - ref).pow(2).mean().sqrt()))
        assert float(wav[b, lens[b]:].abs().sum()) == 0.0
    log_parity(f"[generate() end to end, max_length {L}] {Lout} columns generated, kept frames {[l // DA.DAC_TINY.hop_length for l in lens]}; choices of the first {n_judged} passes outside 1e-4 of the "
               f"oracle's best at the same history: {bad}; waveform rms vs the oracle pipeline on the engine's ids {worst:.2e}", LOG)
    assert bad == 0
    assert worst <= 1e-4, worst
    assert min(lens) // DA.DAC_TINY.hop_length >= 1000  # (random heads emit the 64 special ids now and then: those frames are filtered, :3627-3636)
