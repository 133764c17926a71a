"""GPU parity tests of the decoder-LM engine (HIP kernels behind the C ABI) against the oracle and the golden
vectors generated from the reference's own classes.

Tolerances (stated per the parity bar):
  fp32 engine  vs fp32 oracle / reference logits : |Δlogit| <= 2e-5 (fp32 summation-order noise; logits are O(0.3))
  fp32 engine  greedy token ids                  : bit-exact (margin-safe seeds; min top-2 margin recorded in fixture)
  bf16 engine  vs the SAME bf16-quantised model evaluated by the oracle: |Δlogit| <= 1e-2 (activation re-rounding)
"""
import os

import numpy as np
import pytest
import torch

import cases as C
from conftest import GOLD
from helpers import make_engine, spec_from_gold, t
from oracle import decoder_oracle as DO

pytestmark = pytest.mark.gpu


def _teacher_forced(eng, g, spec, sample_first=False):
    """prefill + N steps feeding fixture ids; returns list of logits [B*K, V] (cpu)."""
    eng.set_gen_params(max_length=16)  # < 2K-1: delay pattern disabled (modeling:246-247), so raw ids are fed as-is
    enc_mask = t(g["enc_mask"]) if "enc_mask" in g else None
    eng.prefill(t(g["enc"]), enc_mask, t(g["prompt"]), t(g["prompt_mask"]) if "prompt_mask" in g else None, sample=False)
    outs = [eng.logits().cpu()]
    for s in range(g["step_ids"].shape[0]):
        eng.push_tokens(t(g["step_ids"][s]).reshape(-1))
        eng.step_forward()
        outs.append(eng.logits().cpu())
    return outs


@pytest.mark.parametrize("variant", ["sin", "rope", "gqa"])
def test_fp32_logits_match_reference_golden(variant):
    g = np.load(os.path.join(GOLD, f"decoder_{variant}.npz"))
    spec = spec_from_gold(g["spec"])
    sd = DO.make_decoder_weights(spec, seed=int(g["weight_seed"]))
    eng = make_engine(spec, sd, torch.float32)
    outs = _teacher_forced(eng, g, spec)
    assert (outs[0] - t(g["prefill_logits"])).abs().max() < 2e-5
    for s in range(g["step_ids"].shape[0]):
        d = (outs[s + 1] - t(g["step_logits"][s])).abs().max()
        assert d < 2e-5, (s, float(d))


@pytest.mark.parametrize("variant", ["sin", "rope", "gqa"])
def test_bf16_logits_match_quantised_oracle(variant):
    g = np.load(os.path.join(GOLD, f"decoder_{variant}.npz"))
    spec = spec_from_gold(g["spec"])
    sd = DO.make_decoder_weights(spec, seed=int(g["weight_seed"]))
    eng = make_engine(spec, sd, torch.bfloat16)
    outs = _teacher_forced(eng, g, spec)
    orc = DO.DecoderOracle(spec, sd, precision="bf16")
    K = spec.num_codebooks
    bsz = g["enc"].shape[0]
    ref = [orc.forward(torch.full((bsz * K, 1), spec.bos_token_id), t(g["enc"]), t(g["enc_mask"]), t(g["prompt"]), t(g["prompt_mask"]))[:, -1]]
    for s in range(g["step_ids"].shape[0]):
        ref.append(orc.forward(t(g["step_ids"][s]))[:, -1])
    for a, b in zip(outs, ref):
        assert (a - b).abs().max() < 1e-2
    # and the bf16 model stays close to the fp32 reference logits (quantisation error, informational bound)
    assert (outs[0] - t(g["prefill_logits"])).abs().max() < 5e-2


@pytest.mark.parametrize("variant", ["sin", "rope"])
def test_fp32_greedy_ids_bit_exact_with_eos_paths(variant):
    """Free-running greedy generation incl. EOS gating, finished-row padding and early stop, vs ids produced by
    the reference forward + reference ParlerTTSLogitsProcessor + reference delay helpers."""
    g = np.load(os.path.join(GOLD, f"greedy_{variant}.npz"))
    spec = spec_from_gold(g["spec"])
    sd = DO.make_decoder_weights(spec, seed=int(g["weight_seed"]))
    for k in range(spec.num_codebooks):
        sd[f"lm_heads.{k}.weight"][spec.eos_token_id] *= float(g["eos_row_gain"])
    eng = make_engine(spec, sd, torch.float32)
    eng.set_gen_params(max_length=int(g["max_length"]), min_new_tokens=int(g["min_new_tokens"]))
    ids = eng.generate_ids(t(g["enc"]), t(g["enc_mask"]), t(g["prompt"]), t(g["prompt_mask"]), poll_every=7).cpu()
    assert torch.equal(ids, t(g["sequences"]))
    # over-running after everything finished must be a no-op (device-side all-done check)
    eng.decode_steps(5)
    cur, fin = eng.state()
    assert fin and cur == g["sequences"].shape[1]


@pytest.mark.parametrize("seed", C.VOICE_LM_SEEDS)
def test_voice_prompt_prefix_continuation_ids_bit_exact(seed):
    """decoder_input_ids prefix (modeling:3136-3194, :205-276): the engine teacher-forces the given code columns one position
    at a time; the oracle (like the reference) runs them in ONE multi-column forward. Same raw ids, incl. the forced
    delayed prompt values the model sees beyond the prefix, MinNewTokens counted from the given length, EOS paths."""
    spec, sd, enc, prompt, pre, gp = C.voice_lm_case(seed)
    ref = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, None, prompt, None, gp, decoder_input_ids=pre)
    assert ref.min_margin >= C.MARGIN  # seeds scanned on the oracle (tools/scan_margin_seeds.py)
    eng = make_engine(spec, sd, torch.float32, max_batch=1)
    eng.set_gen_params(max_length=gp.max_length, min_new_tokens=gp.min_new_tokens)
    ids = eng.generate_ids(enc, None, prompt, None, poll_every=5, audio_prefix=pre[None]).cpu()
    assert torch.equal(ids, ref.sequences)
    # a following call without a prefix must not see the old one
    ref0 = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, None, prompt, None, gp)
    assert ref0.min_margin >= C.MARGIN
    assert torch.equal(eng.generate_ids(enc, None, prompt, None).cpu(), ref0.sequences)


@pytest.mark.parametrize("batched", [True, False])
def test_voice_prompt_batched_multi_column_prefill(batched):
    """The T voice-prompt columns run in the SAME prefill pass as the prompt + BOS column when the engine's row capacity holds
    P + 1 + T positions (what generate() allocates), as the reference's single multi-column forward does (:3136-3194); with a
    smaller capacity they are teacher-forced one position at a time. Both must give the oracle's ids (2 utterances, ragged
    prompt mask, EOS paths)."""
    spec, sd, enc, prompt, pre, gp = C.voice_lm_case(C.VOICE_LM_SEEDS[0])
    T = pre.shape[-1]
    ref = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, None, prompt, None, gp, decoder_input_ids=pre)
    assert ref.min_margin >= C.MARGIN
    P = prompt.shape[1]
    eng = make_engine(spec, sd, torch.float32, max_batch=1, max_prompt=(P + 1 + T) if batched else (P + 1))
    eng.set_gen_params(max_length=gp.max_length, min_new_tokens=gp.min_new_tokens)
    ids = eng.generate_ids(enc, None, prompt, None, poll_every=5, audio_prefix=pre[None]).cpu()
    assert torch.equal(ids, ref.sequences)
    # first-step logits of the continuation (position P + 1 + T) vs the oracle's multi-column forward
    eng.set_audio_prefix(pre[None])
    eng.prefill(enc, None, prompt, None, sample=False)
    orc = DO.DecoderOracle(spec, sd)
    seq0 = torch.cat([torch.full((spec.num_codebooks, 1), spec.bos_token_id), pre], 1)
    delayed, pattern = DO.build_delay_pattern_mask(seq0, spec.bos_token_id, spec.pad_token_id, gp.max_length, spec.num_codebooks)
    lg = orc.forward(DO.apply_delay_pattern_mask(delayed, pattern), enc, None, prompt, None)[:, -1]
    assert (eng.logits().cpu() - lg).abs().max() < 2e-5


def test_early_stop_when_all_rows_hit_eos():
    """Every codebook emits EOS as soon as the gate lets it: the loop must end after min_new + K steps, not at
    max_length. All non-EOS LM-head rows are zero, so blocked rows see an all-equal score vector: also checks the
    first-index tie-break of argmax (token 0)."""
    spec = DO.TINY
    sd = DO.make_decoder_weights(spec, seed=3)
    boost = torch.zeros(spec.hidden_size)
    for k in range(spec.num_codebooks):
        w = sd[f"lm_heads.{k}.weight"]
        eos_row = w[spec.eos_token_id].clone()
        w.zero_()
        w[spec.eos_token_id] = eos_row
        boost += eos_row
    sd["model.decoder.layer_norm.bias"] = sd["model.decoder.layer_norm.bias"] + 40.0 * boost  # EOS logit ~ +2 everywhere
    g = torch.Generator().manual_seed(0)
    enc = torch.randn(2, 5, spec.hidden_size, generator=g)
    gp = DO.GenParams(max_length=64, min_new_tokens=2)
    ref = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, None, None, None, gp)
    assert ref.sequences.shape[1] == 1 + 2 + spec.num_codebooks  # BOS + min_new zeros + one EOS per codebook, staggered
    eng = make_engine(spec, sd, torch.float32)
    eng.set_gen_params(max_length=64, min_new_tokens=2)
    ids = eng.generate_ids(enc, None, None, None, poll_every=4).cpu()
    assert torch.equal(ids, ref.sequences)


@pytest.mark.parametrize("bsz", [1, 3, 20])
def test_batch_sizes_and_two_mfma_tiles(bsz):
    """bsz=20 > 16 exercises the two-accumulator (batch <= 32) GEMM path and ragged masks per row."""
    spec, sd, enc, enc_mask, prompt, prompt_mask, gp = C.batch_case(bsz)
    orc = DO.DecoderOracle(spec, sd)
    ref = DO.sample_loop(orc, enc, enc_mask, prompt, prompt_mask, gp, keep_logits=True)
    assert ref.min_margin >= C.MARGIN  # seeds scanned on the oracle (tools/scan_margin_seeds.py)
    eng = make_engine(spec, sd, torch.float32, max_batch=bsz)
    eng.set_gen_params(max_length=gp.max_length, min_new_tokens=gp.min_new_tokens)
    eng.prefill(enc, enc_mask, prompt, prompt_mask, sample=False)
    assert (eng.logits().cpu() - ref.step_logits[0]).abs().max() < 2e-5
    ids = eng.generate_ids(enc, enc_mask, prompt, prompt_mask).cpu()
    assert torch.equal(ids, ref.sequences)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_prefill_block_gemm_path(dtype):
    """Prefill with > 256 rows (12 utterances x 24 positions; cross K/V over 12 x 40 encoder rows) runs the register-blocked
    GEMM kernel (gemm_block_kernel): first-step logits vs the oracle, ragged masks, then free-running ids."""
    spec, sd, enc, enc_mask, prompt, prompt_mask, gp = C.block_case()
    bsz = enc.shape[0]
    quant = dtype == torch.bfloat16
    orc = DO.DecoderOracle(spec, sd, precision="bf16" if quant else "fp32")  # bf16: the same bf16-quantised model
    ref = DO.sample_loop(orc, enc, enc_mask, prompt, prompt_mask, gp, keep_logits=True)
    eng = make_engine(spec, sd, dtype, max_batch=bsz, max_ctx=64, max_enc=48, max_prompt=32)
    eng.set_gen_params(max_length=gp.max_length, min_new_tokens=gp.min_new_tokens)
    eng.prefill(enc, enc_mask, prompt, prompt_mask, sample=False)
    err = (eng.logits().cpu() - ref.step_logits[0]).abs().max()
    assert err < (1e-2 if quant else 2e-5), float(err)
    if not quant:  # bit-exact ids are an fp32 claim; the seed is margin-safe on the fp32 oracle
        assert ref.min_margin >= C.MARGIN
        assert torch.equal(eng.generate_ids(enc, enc_mask, prompt, prompt_mask).cpu(), ref.sequences)


@pytest.mark.parametrize("bsz", [3, 12])
def test_grouped_query_attention_free_running(bsz):
    """Grouped-query attention (repeat_kv modeling:280-289; K/V projections with fewer heads :449-452): 4 query heads on 2 self /
    1 cross K/V heads, RoPE, ragged masks. bsz 3: fused cross block + fused-prologue GEMMs; bsz 12: prep / plain
    cross-attention path and the block-GEMM prefill. First-step logits and free-running greedy ids vs the oracle (itself pinned
    against the reference's GQA forward, tests/golden/decoder_gqa.npz)."""
    spec, sd, enc, enc_mask, prompt, prompt_mask, gp = C.gqa_case(bsz)
    ref = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, enc_mask, prompt, prompt_mask, gp, keep_logits=True)
    assert ref.min_margin >= C.MARGIN  # seeds scanned on the oracle (tools/scan_margin_seeds.py)
    eng = make_engine(spec, sd, torch.float32, max_batch=bsz, max_ctx=64, max_enc=32, max_prompt=32)
    eng.set_gen_params(max_length=gp.max_length, min_new_tokens=gp.min_new_tokens)
    eng.prefill(enc, enc_mask, prompt, prompt_mask, sample=False)
    assert (eng.logits().cpu() - ref.step_logits[0]).abs().max() < 2e-5
    for s in range(1, 8):  # teacher-forced decode steps on the oracle's own ids: logits comparable whatever the arg-max margins
        eng.push_tokens(ref.sequences[:, s])
        eng.step_forward()
        assert (eng.logits().cpu() - ref.step_logits[s]).abs().max() < 2e-5, s
    assert torch.equal(eng.generate_ids(enc, enc_mask, prompt, prompt_mask).cpu(), ref.sequences)


def test_mini_width_two_layers_fp32_and_bf16():
    """Mini-v1 widths (H=1024, 16 heads, F=4096, V=1088, K=9) with 2 layers: kernel tiling at the real shapes."""
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=1234)
    g = torch.Generator().manual_seed(1)
    bsz, N, P = 2, 24, 9
    enc = torch.randn(bsz, N, spec.hidden_size, generator=g)
    prompt = torch.randn(bsz, P, spec.hidden_size, generator=g) * 0.5
    step_ids = torch.randint(0, 1024, (4, bsz * spec.num_codebooks), generator=g)
    for dtype, prec, tol in ((torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)):
        orc = DO.DecoderOracle(spec, sd, precision=prec)
        ref = [orc.forward(torch.full((bsz * 9, 1), 1025), enc, None, prompt, None)[:, -1]]
        for s in range(4):
            ref.append(orc.forward(step_ids[s][:, None])[:, -1])
        eng = make_engine(spec, sd, dtype, max_batch=bsz, max_ctx=64, max_enc=32, max_prompt=16)
        eng.set_gen_params(max_length=16)
        eng.prefill(enc, None, prompt, None, sample=False)
        outs = [eng.logits().cpu()]
        for s in range(4):
            eng.push_tokens(step_ids[s])
            eng.step_forward()
            outs.append(eng.logits().cpu())
        for a, b in zip(outs, ref):
            assert (a - b).abs().max() < tol, (prec, float((a - b).abs().max()))
        eng.close()


def _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz, N, P, steps, masks, seed, max_ctx=64, weights_fp8=False, oracle_sd=None, max_batch=None,
                              return_logits=False, kv_fp8=False, oracle_kv_fp8=None):
    g = torch.Generator().manual_seed(seed)
    enc = torch.randn(bsz, N, spec.hidden_size, generator=g)
    prompt = torch.randn(bsz, P, spec.hidden_size, generator=g) * 0.5
    enc_mask = prompt_mask = None
    if masks:
        enc_mask, prompt_mask = C.ragged_masks(bsz, N, P, enc_step=2)
        enc_mask[0, N - 3:] = 0  # batch 1 must see padding too
        prompt_mask[0, :2] = 0
        enc = enc * enc_mask[..., None]
    step_ids = torch.randint(0, 1024, (steps, bsz * spec.num_codebooks), generator=g)
    orc = DO.DecoderOracle(spec, oracle_sd if oracle_sd is not None else sd, precision=prec)
    orc.kv_fp8 = kv_fp8 if oracle_kv_fp8 is None else oracle_kv_fp8
    ref = [orc.forward(torch.full((bsz * 9, 1), 1025), enc, enc_mask, prompt, prompt_mask)[:, -1]]
    for s in range(steps):
        ref.append(orc.forward(step_ids[s][:, None])[:, -1])
    eng = make_engine(spec, sd, dtype, max_batch=max_batch or bsz, max_ctx=max_ctx, max_enc=max(N, 16), max_prompt=P + 1, weights_fp8=weights_fp8, kv_fp8=kv_fp8)
    eng.set_gen_params(max_length=16)
    eng.prefill(enc, enc_mask, prompt, prompt_mask, sample=False)
    outs = [eng.logits().cpu()]
    for s in range(steps):
        eng.push_tokens(step_ids[s])
        eng.step_forward()
        outs.append(eng.logits().cpu())
    eng.close()
    if return_logits:
        return outs, ref
    return max(float((a - b).abs().max()) for a, b in zip(outs, ref))


@pytest.mark.parametrize("rope", [False, True])
@pytest.mark.parametrize("gqa", [False, True])
def test_single_utterance_gemv_step_variants(rope, gqa):
    """bs = 1 at Mini width runs the row-per-wave GEMV step (ptts_gemv_kernels.h): LayerNorm / split-KV-combine prologue waves,
    COPY kernels, 1-wave cross-attention. Padded description + prompt masks, RoPE (q rotated, keys not, in the cross block),
    grouped-query attention (N_qkv = 1536: 2 rows per wave), fp32 and bf16, 6 teacher-forced steps vs the oracle."""
    kw = dict(num_hidden_layers=2, max_position_embeddings=512, rope_embeddings=rope)
    if gqa:
        kw.update(num_key_value_heads=4, num_cross_attention_key_value_heads=2)
    spec = DO.DecoderSpec(**kw)
    sd = DO.make_decoder_weights(spec, seed=41)
    for dtype, prec, tol in ((torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)):
        err = _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz=1, N=21, P=6, steps=6, masks=True, seed=8)
        assert err < tol, (prec, err)


def test_single_utterance_gemv_step_long_context_all_split_counts():
    """GEMV step with the KV length growing through several row-group batches: max_ctx 1100 -> 4 KV splits (combine prologue
    S = 4); max_ctx 300 -> 2 splits; fp32, 40 free-running columns compared by logits on the oracle's ids."""
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=2048)
    sd = DO.make_decoder_weights(spec, seed=43)
    for max_ctx in (300, 1100):
        err = _teacher_forced_vs_oracle(spec, sd, torch.float32, "fp32", bsz=1, N=9, P=200, steps=40, masks=False, seed=3, max_ctx=max_ctx)
        assert err < 5e-5, (max_ctx, err)


@pytest.mark.parametrize("gqa", [False, True])
def test_fused_qkv_attention_node_single_utterance(gqa, monkeypatch):
    """qkv_attn_kernel (single utterance, sinusoidal positions): LN1 + the head's q / k / v rows + split-KV self-attention + append as ONE
    node, the new position as a slot of its own in the combine prologue (GV_ATTN2). fp32 at H = 512 and H = 1024: the fused
    step against the two-node step (PTTS_NO_FUSE_QA=1) within fp32 summation noise AND both against the oracle; short context with a padded
    prompt (2 KV splits) and a 1100-position prompt (4 splits, second K/V batch of the attention loop); grouped-query attention (one
    writer per K/V group). bf16 at Mini and Large width (NCH 2 / 3) against the bf16 oracle."""
    kw = dict(hidden_size=512, num_attention_heads=8, ffn_dim=1024, num_hidden_layers=3, max_position_embeddings=2048)
    if gqa:
        kw.update(num_key_value_heads=2, num_cross_attention_key_value_heads=2)
    spec = DO.DecoderSpec(**kw)
    sd = DO.make_decoder_weights(spec, seed=67)
    for P, max_ctx, masks, steps in ((6, 300, True, 6), (1100, 1400, False, 4)):
        runs = {}
        for fuse in (True, False):
            if fuse:
                monkeypatch.delenv("PTTS_NO_FUSE_QA", raising=False)
            else:
                monkeypatch.setenv("PTTS_NO_FUSE_QA", "1")
            runs[fuse], ref = _teacher_forced_vs_oracle(spec, sd, torch.float32, "fp32", bsz=1, N=21, P=P, steps=steps, masks=masks, seed=5, max_ctx=max_ctx,
                                                        return_logits=True)
        monkeypatch.delenv("PTTS_NO_FUSE_QA", raising=False)
        ab = max(float((a - b).abs().max()) for a, b in zip(runs[True], runs[False]))
        assert ab < 2e-5, (P, "fused vs two nodes", ab)
        assert ab > 0.0, "the fused node did not run (identical logits: same kernels on both sides)"
        for fuse in (True, False):
            err = max(float((a - b).abs().max()) for a, b in zip(runs[fuse], ref))
            assert err < 5e-5, (P, fuse, err)
    if not gqa:
        for kw2, seed, tol in ((dict(), 71, 2e-2), (dict(hidden_size=1536, num_attention_heads=24, ffn_dim=6144), 73, 3e-2)):
            spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=2048, **kw2)
            sd = DO.make_decoder_weights(spec, seed=seed)
            err = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", bsz=1, N=21, P=1100, steps=4, masks=False, seed=6, max_ctx=1400)
            assert err < tol, (kw2, err)
        # fp32 at Mini-v1 width (the bit-exact parity engine of bench.py's fp32_parity_mode): 8 weight rows x 4 KB per wave, U = 4
        spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=2048)
        sd = DO.make_decoder_weights(spec, seed=97)
        for P, max_ctx, masks in ((7, 200, True), (700, 900, False)):
            runs = {}
            for fuse in (True, False):
                monkeypatch.setenv("PTTS_NO_FUSE_QA", "0" if fuse else "1")
                runs[fuse], ref = _teacher_forced_vs_oracle(spec, sd, torch.float32, "fp32", bsz=1, N=33, P=P, steps=4, masks=masks, seed=9, max_ctx=max_ctx,
                                                            return_logits=True)
            monkeypatch.delenv("PTTS_NO_FUSE_QA", raising=False)
            ab = max(float((a - b).abs().max()) for a, b in zip(runs[True], runs[False]))
            assert 0.0 < ab < 2e-5, (P, "fp32 Mini width, fused vs two nodes", ab)
            for fuse in (True, False):
                err = max(float((a - b).abs().max()) for a, b in zip(runs[fuse], ref))
                assert err < 5e-5, (P, fuse, err)


@pytest.mark.parametrize("gqa", [False, True])
@pytest.mark.parametrize("bsz", [2, 3, 8])
def test_fused_qkv_attention_node_two_to_eight_utterances(bsz, gqa, monkeypatch):
    """qkv_attn_kernel with one grid slice per utterance (round 5): 2..8 utterances can run LN1 + q / k / v rows + split-KV self-attention +
    append as ONE node, combined per utterance by the out_proj node's GV_ATTN2 prologue (instances for 2..4 and 5..8 utterances). Default: up
    to 3 utterances (measured); PTTS_FUSE_QA_MAX=8 here so the 5..8 instances are covered too. bf16 (the fp32 engine serves one utterance on
    this path): the fused step against the two-node step (PTTS_FUSE_QA_MAX=1: one utterance only) AND both against the bf16 oracle, ragged description / prompt
    masks, a short context (1 split) and a 600-position prompt (4 splits at <= 4 utterances, 2 above: second K/V batch of the attention
    loop), grouped-query attention (one writer per K/V group and utterance). That the fused node really is in the step is read off the
    captured graph: one kernel node less per layer (ptts_debug_graph_nodes) - in bf16 the two variants can agree to the last bit."""
    kw = dict(num_hidden_layers=2, max_position_embeddings=2048)
    if gqa:
        kw.update(num_key_value_heads=4, num_cross_attention_key_value_heads=2)
    spec = DO.DecoderSpec(**kw)
    sd = DO.make_decoder_weights(spec, seed=83)
    for P, max_ctx, steps in ((6, 200, 5), (600, 800, 3)):
        runs = {}
        for fuse in (True, False):
            monkeypatch.setenv("PTTS_FUSE_QA_MAX", "8" if fuse else "1")
            runs[fuse], ref = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", bsz=bsz, N=21, P=P, steps=steps, masks=True, seed=11 + bsz, max_ctx=max_ctx,
                                                        return_logits=True)
        ab = max(float((a - b).abs().max()) for a, b in zip(runs[True], runs[False]))
        assert ab < 2e-2, (bsz, P, "fused vs two nodes", ab)
        for fuse in (True, False):
            err = max(float((a - b).abs().max()) for a, b in zip(runs[fuse], ref))
            assert err < 2e-2, (bsz, P, fuse, err)
    # node count of the captured step: 6 nodes per layer fused (qkv_attn, combine + out_proj, xq_attn, cross out_proj, LN3 + fc1, fc2), 7 otherwise
    g = torch.Generator().manual_seed(1)
    enc, prompt = torch.randn(bsz, 9, spec.hidden_size, generator=g), torch.randn(bsz, 4, spec.hidden_size, generator=g)
    nodes = {}
    for fuse in (True, False):
        monkeypatch.setenv("PTTS_FUSE_QA_MAX", "8" if fuse else "1")
        eng = make_engine(spec, sd, torch.bfloat16, max_batch=bsz, max_ctx=64, max_enc=16, max_prompt=5)
        eng.set_gen_params(max_length=12, min_new_tokens=11)
        eng.prefill(enc, None, prompt, None, sample=True)
        eng.decode_steps(2)
        nodes[fuse] = eng.graph_nodes()
        eng.close()
    assert nodes[False] == 7 * spec.num_hidden_layers + 2 and nodes[True] == 6 * spec.num_hidden_layers + 2, nodes


def test_fused_qkv_attention_default_bound_is_three_utterances():
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=256)
    sd = DO.make_decoder_weights(spec, seed=5)
    g = torch.Generator().manual_seed(2)
    for bsz, per_layer in ((1, 5), (2, 6), (3, 6), (4, 7)):
        enc, prompt = torch.randn(bsz, 9, spec.hidden_size, generator=g), torch.randn(bsz, 4, spec.hidden_size, generator=g)
        eng = make_engine(spec, sd, torch.bfloat16, max_batch=bsz, max_ctx=64, max_enc=16, max_prompt=5)
        eng.set_gen_params(max_length=12, min_new_tokens=11)
        eng.prefill(enc, None, prompt, None, sample=True)
        eng.decode_steps(2)
        assert eng.graph_nodes() == per_layer * spec.num_hidden_layers + 2, (bsz, eng.graph_nodes())
        eng.close()


@pytest.mark.parametrize("rope", [False, True])
@pytest.mark.parametrize("P", [70, 32])
def test_prefill_attention_tiled_kernel_vs_one_workgroup_per_row(rope, P, monkeypatch):
    """The three prefill attentions against the oracle and against each other: prefill_attn_mfma_kernel (round 6: exact-f32 MFMA, up to 64 query
    rows per workgroup; PTTS_PREFILL_ATTN=2, the default from 128 (utterance, head) pairs up), prefill_attn_kernel (round 5: 8 query rows per
    workgroup share the K / V tile; =1) and attn_kernel's one-workgroup-per-row prefill (=0). 70 prompt positions = 71 query rows (two 64-query
    workgroups, two key blocks, causal boundary inside a block) and 32 = 33 rows (bench.py's prefill: three waves, the last with one query);
    ragged prompt / description masks incl. a fully padded prompt head, grouped-query attention, RoPE (q rotated in both blocks, cross keys
    not), fp32 and bf16; 3 utterances (strip GEMMs, fragment-order activations) and 9 (row-major activations, the > 256-row GEMMs)."""
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512, rope_embeddings=rope, num_key_value_heads=4, num_cross_attention_key_value_heads=2)
    sd = DO.make_decoder_weights(spec, seed=53)
    for bsz in (3, 9):
        for dtype, prec, tol in ((torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)):
            runs = {}
            for mode in ("2", "1", "0"):
                monkeypatch.setenv("PTTS_PREFILL_ATTN", mode)
                runs[mode], ref = _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz=bsz, N=70, P=P, steps=1, masks=True, seed=17, max_ctx=128, return_logits=True)
            monkeypatch.delenv("PTTS_PREFILL_ATTN", raising=False)
            for mode in ("2", "1", "0"):
                err = max(float((a - b).abs().max()) for a, b in zip(runs[mode], ref))
                assert err < tol, (bsz, prec, {"2": "f32 MFMA", "1": "tiled", "0": "per row"}[mode], err)
            if dtype == torch.float32:
                for m in ("2", "1"):
                    ab = max(float((a - b).abs().max()) for a, b in zip(runs[m], runs["0"]))
                    assert 0.0 < ab < 2e-5, (bsz, m, "vs per-row prefill attention", ab)


@pytest.mark.parametrize("bsz", [12, 40, 70])
def test_e4m3_kv_cache_mode(bsz):
    """ptts_config::kv_fp8 (opt-in, engines of more than 8 utterances): the self-attention cache holds e4m3 rows + one power-of-two scale per
    (utterance, head, position), quantised at append - by kv_append_kernel for the prefill rows, by attn_kernel's fused append at decode.
    Teacher-forced logits against the bf16 oracle applying the SAME quantiser (oracle/fp8_oracle.py: quantize_kv_rows) within the bf16
    tolerance; the mode is really on (logits differ from the bf16-cache engine) and the quantiser is the right one (the engine is closer to the
    quantised oracle than to the plain one). Ragged masks; 70 utterances = the 16-row LayerNorm + projection nodes, exact-length K/V fetch."""
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=29)
    kw = dict(bsz=bsz, N=13, P=5, steps=4, masks=True, seed=7, max_ctx=64, return_logits=True)
    q_out, q_ref = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", kv_fp8=True, **kw)
    p_out, p_ref = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", kv_fp8=False, **kw)
    err_q = max(float((a - b).abs().max()) for a, b in zip(q_out, q_ref))
    err_cross = max(float((a - b).abs().max()) for a, b in zip(q_out, p_ref))
    on = max(float((a - b).abs().max()) for a, b in zip(q_out, p_out))
    model = max(float((a - b).abs().max()) for a, b in zip(q_ref, p_ref))
    from helpers import log_parity

    log_parity(f"[e4m3 KV cache, {bsz} utterances] max |dlogit| vs the quantised-cache oracle {err_q:.2e}; vs the bf16-cache oracle {err_cross:.2e}; "
               f"engine kv8 vs engine bf16 cache {on:.2e}; oracle kv8 vs oracle bf16 cache {model:.2e}", "r05_parity_kv8.txt")
    assert err_q < 2e-2, err_q
    assert on > 1e-3, "kv_fp8 engine produced the bf16-cache logits: the mode is not on"
    assert err_q < err_cross, (err_q, err_cross)


def test_e4m3_kv_cache_needs_the_wide_bf16_engine():
    spec = DO.DecoderSpec(num_hidden_layers=1, max_position_embeddings=128)
    sd = DO.make_decoder_weights(spec, seed=1)
    with pytest.raises(NotImplementedError, match="kv_fp8"):
        make_engine(spec, sd, torch.bfloat16, max_batch=4, kv_fp8=True)
    with pytest.raises(NotImplementedError, match="kv_fp8"):
        make_engine(spec, sd, torch.float32, max_batch=16, kv_fp8=True)


@pytest.mark.parametrize("gqa", [False, True])
def test_fused_cross_block_node_single_utterance(gqa, monkeypatch):
    """xfold_attn_kernel (single utterance, folded cross block): LN2 + the head's rows of M + per-head softmax + the head's columns of U as ONE
    node of per-head partial rows, summed in row order by the LN3 + fc1 node's prologue (GV_LNP), whose workgroup 0 also publishes the
    summed row as the fc2 node's residual operand. fp32 at H = 512: the fused step (PTTS_FUSE_X=1) against the two-node step (PTTS_FUSE_X=0) within
    fp32 summation noise AND both against the oracle, with a padded description (mask + fewer than 64 valid positions) and without, both
    workgroup shapes (PTTS_FUSE_X_NUR = 2 / 4), grouped cross K/V heads; bf16 at Mini and Large width against the bf16 oracle (e4m3 engines
    take the same node by default: their single-utterance tests below run it)."""
    kw = dict(hidden_size=512, num_attention_heads=8, ffn_dim=1024, num_hidden_layers=3, max_position_embeddings=512)
    if gqa:
        kw.update(num_key_value_heads=2, num_cross_attention_key_value_heads=2)
    spec = DO.DecoderSpec(**kw)
    sd = DO.make_decoder_weights(spec, seed=79)
    for N, masks, nur in ((21, True, 2), (64, False, 4), (37, True, 4)):
        runs = {}
        monkeypatch.setenv("PTTS_FUSE_X_NUR", str(nur))
        for fuse in (True, False):
            monkeypatch.setenv("PTTS_FUSE_X", "1" if fuse else "0")
            runs[fuse], ref = _teacher_forced_vs_oracle(spec, sd, torch.float32, "fp32", bsz=1, N=N, P=6, steps=5, masks=masks, seed=7, max_ctx=64,
                                                        return_logits=True)
        monkeypatch.delenv("PTTS_FUSE_X", raising=False)
        ab = max(float((a - b).abs().max()) for a, b in zip(runs[True], runs[False]))
        assert ab < 2e-5, (N, "fused vs two nodes", ab)
        assert ab > 0.0, "the fused node did not run (identical logits: same kernels on both sides)"
        for fuse in (True, False):
            err = max(float((a - b).abs().max()) for a, b in zip(runs[fuse], ref))
            assert err < 5e-5, (N, fuse, err)
    monkeypatch.delenv("PTTS_FUSE_X_NUR", raising=False)
    if not gqa:
        for kw2, seed, tol in ((dict(), 83, 2e-2), (dict(hidden_size=1536, num_attention_heads=24, ffn_dim=6144), 89, 3e-2)):
            spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512, **kw2)
            sd = DO.make_decoder_weights(spec, seed=seed)
            monkeypatch.setenv("PTTS_FUSE_X", "1")  # off by default above hidden 1024
            for nur in (2, 4):
                monkeypatch.setenv("PTTS_FUSE_X_NUR", str(nur))
                err = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", bsz=1, N=45, P=9, steps=4, masks=True, seed=8, max_ctx=64)
                assert err < tol, (kw2, nur, err)
            monkeypatch.delenv("PTTS_FUSE_X_NUR", raising=False)
            monkeypatch.delenv("PTTS_FUSE_X", raising=False)


@pytest.mark.parametrize("bsz,gqa", [(1, False), (2, False), (3, True), (8, False)])
def test_fused_cross_q_attention_node_gemv_step(bsz, gqa, monkeypatch):
    """xq_attn_kernel (1..8 utterances on the GEMV step, no static fold): LN2 + the head's cross-q rows + cross-attention of one (head,
    utterance) as ONE node. fp32 at H = 512: fused against the two-node path (PTTS_NO_FUSE_XQ=1) within fp32 summation noise and both
    against the oracle - ragged description masks, a description longer than one K/V batch per wave (N = 300: the attention loop's
    second batch) and longer than the fold's 64 positions at one utterance; grouped cross K/V heads. bf16 at Mini / Large width against
    the bf16 oracle. (fp32 engines are built for one utterance: the batched cases run on the bf16 engine only.)"""
    kw = dict(hidden_size=512, num_attention_heads=8, ffn_dim=1024, num_hidden_layers=3, max_position_embeddings=512)
    if gqa:
        kw.update(num_key_value_heads=2, num_cross_attention_key_value_heads=2)
    spec = DO.DecoderSpec(**kw)
    sd = DO.make_decoder_weights(spec, seed=101)
    if bsz == 1:
        for N, masks in ((80, True), (300, False)):
            runs = {}
            for fuse in (True, False):
                monkeypatch.setenv("PTTS_NO_FUSE_XQ", "0" if fuse else "1")
                runs[fuse], ref = _teacher_forced_vs_oracle(spec, sd, torch.float32, "fp32", bsz=1, N=N, P=6, steps=4, masks=masks, seed=11, max_ctx=64,
                                                            return_logits=True)
            monkeypatch.delenv("PTTS_NO_FUSE_XQ", raising=False)
            ab = max(float((a - b).abs().max()) for a, b in zip(runs[True], runs[False]))
            assert 0.0 < ab < 2e-5, (N, "fused vs two nodes", ab)
            for fuse in (True, False):
                err = max(float((a - b).abs().max()) for a, b in zip(runs[fuse], ref))
                assert err < 5e-5, (N, fuse, err)
    for kw2, seed, tol in ((dict(), 103, 2e-2), (dict(hidden_size=1536, num_attention_heads=24, ffn_dim=6144), 107, 3e-2)):
        if gqa:
            kw2 = dict(kw2, num_key_value_heads=4, num_cross_attention_key_value_heads=4)
        spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512, **kw2)
        sd = DO.make_decoder_weights(spec, seed=seed)
        errs = {}
        for fuse in (True, False):
            monkeypatch.setenv("PTTS_NO_FUSE_XQ", "0" if fuse else "1")
            errs[fuse] = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", bsz=bsz, N=70, P=9, steps=3, masks=True, seed=12, max_ctx=64)
        monkeypatch.delenv("PTTS_NO_FUSE_XQ", raising=False)
        assert errs[True] < tol and errs[False] < tol, (kw2, errs)


@pytest.mark.parametrize("bsz", [2, 3, 4, 5, 6, 8])
def test_gemv_step_batch_2_to_8(bsz):
    """Batch 2..8 on the bf16 engine runs the GEMV step with every weight row read once for all utterances (MB = 4 / 8 instances:
    one prologue wave per utterance, M dot products per weight chunk; at 5..8 the K = 4096 activation rows of fc2 are staged in LDS
    once per workgroup and pass through the registers in two groups of 4 utterances, groups past the live batch are skipped); the
    fp32 parity engine keeps the MFMA strip path there. Ragged masks per utterance, 5 teacher-forced steps vs the oracle."""
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=53)
    for dtype, prec, tol in ((torch.bfloat16, "bf16", 2e-2), (torch.float32, "fp32", 5e-5)):
        err = _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz=bsz, N=21, P=6, steps=5, masks=True, seed=10 + bsz)
        assert err < tol, (bsz, prec, err)


@pytest.mark.parametrize("bsz", [5, 8])
def test_gemv_step_large_width_batch_5_and_8(bsz):
    """Large-v1 width (H 1536, ffn 6144) at 5..8 utterances: the K = 6144 activation rows of fc2 (12 chunks per lane per utterance)
    pass through the registers in four groups of 2 utterances (96 KiB would not fit the un-opted dynamic LDS; staging them measured
    no gain there), the K = 1536 nodes in one group of 8."""
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512, hidden_size=1536, num_attention_heads=24, ffn_dim=6144)
    sd = DO.make_decoder_weights(spec, seed=61)
    err = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", bsz=bsz, N=21, P=6, steps=4, masks=True, seed=40 + bsz)
    assert err < 2e-2, (bsz, err)


@pytest.mark.parametrize("width,bsz", [("mini", 1), ("mini", 3), ("mini", 8), ("mini", 12), ("mini", 32), ("mini", 40), ("large", 1), ("large", 4),
                                       ("large", 8), ("large", 12)])
def test_fp8_weight_mode_matches_the_quantised_oracle(width, bsz):
    """weights_fp8 (BASELINE configs[4]): the engine quantises the projection matrices itself (e4m3, power-of-two row scales);
    the oracle evaluates the SAME quantised model (oracle/fp8_oracle.py, hand-rounded e4m3) with its bf16 arithmetic. Batch 1 /
    3 / 4: GEMV step streaming the 1-byte weights (v_cvt_pk_f32_fp8 + fma). Batch >= 5: MFMA strips streaming e4m3 fragment
    pairs converted to bf16 in registers (v_cvt_scalef32_pk_bf16_fp8), row scales on the fp32 accumulators - fused prologues at
    8, prepared rows / producer statistics at 12 and 32 (1 and 2 row tiles), 64-row passes at 40; the cross q projection inside
    the fused cross-block kernel and the prefill read the exact bf16 dequantisation. Same tolerance as the bf16 mode: the two
    sides differ by summation order only."""
    from oracle import fp8_oracle as FO

    kw = dict(num_hidden_layers=2, max_position_embeddings=512)
    if width == "large":
        kw.update(hidden_size=1536, num_attention_heads=24, ffn_dim=6144)
    spec = DO.DecoderSpec(**kw)
    sd = DO.make_decoder_weights(spec, seed=59)
    qsd = FO.quantize_decoder_weights(sd)
    err = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", bsz=bsz, N=21, P=6, steps=4, masks=True, seed=20 + bsz,
                                    weights_fp8=True, oracle_sd=qsd)
    assert err < 2e-2, (width, bsz, err)
    # and quantisation really happened: the un-quantised bf16 model is measurably different from the fp8 one
    err_unq = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", bsz=bsz, N=21, P=6, steps=1, masks=True, seed=20 + bsz,
                                        weights_fp8=True, oracle_sd=None)
    assert err_unq > err


@pytest.mark.parametrize("bsz,fp8", [(3, False), (6, False), (6, True), (8, True)])
def test_mfma_strips_with_fused_prologues_on_a_wide_engine(bsz, fp8):
    """An engine created for 12 utterances (no row-major GEMV copies) called with 3 / 6 / 8: the decode step runs the MFMA strips with
    the fused LayerNorm / split-KV-combine prologues and the fused cross block (the batch <= 8 path of engines that cannot take the
    GEMV step); with weights_fp8 the strips stream e4m3 fragment pairs there too."""
    from oracle import fp8_oracle as FO

    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=67)
    qsd = FO.quantize_decoder_weights(sd) if fp8 else None
    err = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", bsz=bsz, N=21, P=6, steps=4, masks=True, seed=30 + bsz, weights_fp8=fp8,
                                    oracle_sd=qsd, max_batch=12)
    assert err < 2e-2, (bsz, fp8, err)
    if not fp8:
        err = _teacher_forced_vs_oracle(spec, sd, torch.float32, "fp32", bsz=bsz, N=21, P=6, steps=4, masks=True, seed=30 + bsz, max_batch=12)
        assert err < 5e-5, (bsz, err)


@pytest.mark.parametrize("bsz", [12, 32])
def test_mini_width_batch_12_and_32_producer_statistics_layernorm(bsz):
    """8 < batch <= 32 at Mini width: LN2 / LN3 take their row statistics from the producing out_proj GEMM (EPI_RESID strip
    partials -> PRO_LNS prologue, no rows_prep node), fc2 un-split with the residual in its epilogue (round 6). Ragged masks."""
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=47)
    for dtype, prec, tol in ((torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)):
        err = _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz=bsz, N=21, P=6, steps=4, masks=True, seed=bsz)
        assert err < tol, (bsz, prec, err)


@pytest.mark.parametrize("bsz", [9, 12, 32])
@pytest.mark.parametrize("grouped", [True, False])
def test_fused_cross_block_in_groups_of_eight(bsz, grouped, monkeypatch):
    """Batch 9..32: the cross block's LN2 + q projection + cross-attention run in the batch <= 8 fused kernel, one workgroup per (head,
    group of 8 utterances); ragged last group at 9 and 12. PTTS_NO_XATTN_GROUPS=1 (read at engine creation) keeps the two-node path
    (producer-statistics LayerNorm GEMM + attention kernel) alive for A/B. Default tolerances."""
    if not grouped:
        monkeypatch.setenv("PTTS_NO_XATTN_GROUPS", "1")
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=49)
    for dtype, prec, tol in ((torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)):
        err = _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz=bsz, N=21, P=6, steps=4, masks=True, seed=bsz)
        assert err < tol, (bsz, prec, err)


@pytest.mark.parametrize("g", [2, 4])
@pytest.mark.parametrize("bsz,N", [(9, 21), (13, 70), (32, 64)])
def test_fused_cross_block_with_fewer_utterances_per_workgroup(g, bsz, N, monkeypatch):
    """PTTS_XATTN_G = 4 / 2 (read at engine creation): above 8 utterances the fused cross block runs heads x ceil(B / g) workgroups (128 / 256
    at batch 32 instead of 64); the 8 / g waves of an utterance split the description's row groups and merge their partial softmaxes through
    LDS. Ragged last group, ragged masks, a description longer than one batch of row groups (N = 70), Mini and Large widths."""
    monkeypatch.setenv("PTTS_XATTN_G", str(g))
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=49)
    for dtype, prec, tol in ((torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)):
        err = _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz=bsz, N=N, P=6, steps=3, masks=True, seed=bsz)
        assert err < tol, (g, bsz, prec, err)
    if bsz == 13:
        spec = DO.DecoderSpec(hidden_size=1536, num_attention_heads=24, ffn_dim=6144, num_hidden_layers=2, max_position_embeddings=256)
        sd = DO.make_decoder_weights(spec, seed=77)
        err = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", bsz=bsz, N=N, P=5, steps=3, masks=True, seed=3)
        assert err < 3e-2, (g, "large", err)


@pytest.mark.parametrize("mode,g", [(1, 8), (2, 8), (3, 8), (3, 4), (3, 16)])
@pytest.mark.parametrize("bsz", [9, 13, 32, 44, 128])
def test_layernorm_plus_projection_as_one_node(mode, g, bsz, monkeypatch):
    """PTTS_LNPROJ (read at engine creation): above 8 utterances LN1 + QKV (mode >= 1) and LN3 + fc1 + GELU (mode 2: above 32 utterances, mode 3: always)
    run as ONE node tiled over 64 weight rows x g utterances (lnproj_fused_kernel) instead of rows_prep + strip GEMM (fc2 runs un-split with the
    residual in its own epilogue since round 6: no partial rows are folded by these nodes any more). 3 layers, ragged groups
    (9, 13, 44), ragged masks; Mini widths in both dtypes, Large widths in bf16."""
    monkeypatch.setenv("PTTS_LNPROJ", str(mode))
    monkeypatch.setenv("PTTS_LNPROJ_G", str(g))
    spec = DO.DecoderSpec(num_hidden_layers=3, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=53)
    for dtype, prec, tol in ((torch.float32, "fp32", 5e-5), (torch.bfloat16, "bf16", 2e-2)):
        err = _teacher_forced_vs_oracle(spec, sd, dtype, prec, bsz=bsz, N=21, P=6, steps=3, masks=True, seed=bsz)
        assert err < tol, (mode, g, bsz, prec, err)
    if bsz == 13:
        spec = DO.DecoderSpec(hidden_size=1536, num_attention_heads=24, ffn_dim=6144, num_hidden_layers=2, max_position_embeddings=256)
        sd = DO.make_decoder_weights(spec, seed=77)
        err = _teacher_forced_vs_oracle(spec, sd, torch.bfloat16, "bf16", bsz=bsz, N=10, P=5, steps=3, masks=True, seed=3)
        assert err < 3e-2, (mode, g, "large", err)


@pytest.mark.parametrize("bsz", [44, 128])
def test_default_policy_fp32_above_40_utterances(bsz):
    """above 40 utterances the DEFAULT policy runs the 16-row LayerNorm + projection nodes, whose fp32 instances ask for more than
    64 KiB of dynamic LDS (72 KiB at H = 1024, 104 KiB at H = 1536: launch_lnproj's hipFuncSetAttribute opt-in). No PTTS_* switch set: fp32 engine
    at Mini widths (44 and 128 utterances) and at Large widths (44), teacher-forced against the fp32 oracle."""
    spec = DO.DecoderSpec(num_hidden_layers=2, max_position_embeddings=512)
    sd = DO.make_decoder_weights(spec, seed=61)
    err = _teacher_forced_vs_oracle(spec, sd, torch.float32, "fp32", bsz=bsz, N=17, P=5, steps=3, masks=True, seed=bsz)
    assert err < 5e-5, (bsz, err)
    if bsz == 44:
        spec = DO.DecoderSpec(hidden_size=1536, num_attention_heads=24, ffn_dim=6144, num_hidden_layers=2, max_position_embeddings=256)
        sd = DO.make_decoder_weights(spec, seed=77)
        err = _teacher_forced_vs_oracle(spec, sd, torch.float32, "fp32", bsz=bsz, N=10, P=5, steps=3, masks=True, seed=3)
        assert err < 6e-5, ("large", err)


def test_large_v1_width_two_layers_bf16_and_fp32_batch():
    """Large-v1 widths (H=1536, 24 heads, F=6144; init_large_model.py:25-43) with 2 layers, batch 1 and 12:
    6-float4 LayerNorm rows, 6 / 12-wave K splits, the prep-kernel (M > 8) path."""
    spec = DO.DecoderSpec(hidden_size=1536, num_attention_heads=24, ffn_dim=6144, num_hidden_layers=2, max_position_embeddings=256)
    sd = DO.make_decoder_weights(spec, seed=77)
    g = torch.Generator().manual_seed(4)
    for bsz in (1, 12):
        N, P = 10, 5
        enc = torch.randn(bsz, N, spec.hidden_size, generator=g)
        prompt = torch.randn(bsz, P, spec.hidden_size, generator=g) * 0.5
        step_ids = torch.randint(0, 1024, (3, bsz * spec.num_codebooks), generator=g)
        for dtype, prec, tol in ((torch.float32, "fp32", 6e-5), (torch.bfloat16, "bf16", 3e-2)):
            orc = DO.DecoderOracle(spec, sd, precision=prec)
            ref = [orc.forward(torch.full((bsz * 9, 1), 1025), enc, None, prompt, None)[:, -1]]
            for s in range(3):
                ref.append(orc.forward(step_ids[s][:, None])[:, -1])
            eng = make_engine(spec, sd, dtype, max_batch=bsz, max_ctx=64, max_enc=16, max_prompt=8)
            eng.set_gen_params(max_length=16)
            eng.prefill(enc, None, prompt, None, sample=False)
            outs = [eng.logits().cpu()]
            for s in range(3):
                eng.push_tokens(step_ids[s])
                eng.step_forward()
                outs.append(eng.logits().cpu())
            for a, b in zip(outs, ref):
                assert (a - b).abs().max() < tol, (bsz, prec, float((a - b).abs().max()))
            eng.close()


@pytest.mark.parametrize("rope", [False, True])
def test_fused_cross_attention_block_masks_and_rope(rope):
    """Decode at batch <= 8 and Mini width runs LN2 + cross-q + cross-attention as ONE head-parallel kernel: check it
    with a padded description mask per row, a padded prompt, RoPE on/off (q rotated, keys not) and batch 3."""
    spec = DO.DecoderSpec(num_hidden_layers=1, max_position_embeddings=128, rope_embeddings=rope)
    sd = DO.make_decoder_weights(spec, seed=31)
    g = torch.Generator().manual_seed(6)
    bsz, N, P = 3, 21, 6
    enc = torch.randn(bsz, N, spec.hidden_size, generator=g)
    enc_mask = torch.ones(bsz, N, dtype=torch.long)
    enc_mask[1, 13:] = 0
    enc_mask[2, 20:] = 0
    enc = enc * enc_mask[..., None]
    prompt = torch.randn(bsz, P, spec.hidden_size, generator=g) * 0.5
    prompt_mask = torch.ones(bsz, P, dtype=torch.long)
    prompt_mask[2, :3] = 0
    step_ids = torch.randint(0, 1024, (5, bsz * spec.num_codebooks), generator=g)
    orc = DO.DecoderOracle(spec, sd)
    ref = [orc.forward(torch.full((bsz * 9, 1), 1025), enc, enc_mask, prompt, prompt_mask)[:, -1]]
    for s_ in range(5):
        ref.append(orc.forward(step_ids[s_][:, None])[:, -1])
    eng = make_engine(spec, sd, torch.float32, max_batch=bsz, max_ctx=64, max_enc=32, max_prompt=8)
    eng.set_gen_params(max_length=16)
    eng.prefill(enc, enc_mask, prompt, prompt_mask, sample=False)
    outs = [eng.logits().cpu()]
    for s_ in range(5):
        eng.push_tokens(step_ids[s_])
        eng.step_forward()
        outs.append(eng.logits().cpu())
    for a, b in zip(outs, ref):
        assert (a - b).abs().max() < 5e-5, float((a - b).abs().max())


def test_full_mini_v1_fp32_greedy_ids_bit_exact():
    """BASELINE.json configs[0] numerics at the FULL Mini-v1 shape (24 layers, H=1024, F=4096, K=9, V=1088): greedy token
    ids of the fp32 engine vs the oracle's CPU fp32 path, free-running from the same description / prompt tensors: ALL 33
    columns bit-exact. The input seed was scanned on the oracle (min top-2 margin over the run 1.45e-3 against ~3e-6 of fp32
    summation-order noise); the margin is asserted, not used as a guard."""
    spec = DO.MINI_V1
    sd = DO.make_decoder_weights(spec, seed=1234)
    g = torch.Generator().manual_seed(4)
    N, P, L = 16, 8, 33
    enc = torch.randn(1, N, spec.hidden_size, generator=g)
    prompt = torch.randn(1, P, spec.hidden_size, generator=g) * 0.5
    orc = DO.DecoderOracle(spec, sd)
    gp = DO.GenParams(max_length=L, min_new_tokens=L - 1)
    ref = DO.sample_loop(orc, enc, None, prompt, None, gp, keep_logits=True)
    assert ref.min_margin >= 5 * C.MARGIN, ref.min_margin
    eng = make_engine(spec, sd, torch.float32, max_batch=1, max_ctx=64, max_enc=16, max_prompt=16)
    eng.set_gen_params(max_length=L, min_new_tokens=L - 1)
    eng.prefill(enc, None, prompt, None, sample=False)
    assert (eng.logits().cpu() - ref.step_logits[0]).abs().max() < 5e-5
    ids = eng.generate_ids(enc, None, prompt, None).cpu()
    assert ids.shape == ref.sequences.shape == (spec.num_codebooks, L)
    assert torch.equal(ids, ref.sequences)


def test_fused_lm_heads_checkpoint_layout():
    """use_fused_lm_heads (modeling:1834-1840): one [K*V, H] `lm_heads.weight` instead of K `lm_heads.k.weight`."""
    spec = DO.TINY
    sd = DO.make_decoder_weights(spec, seed=9)
    fused = {k: v for k, v in sd.items() if not k.startswith("lm_heads.")}
    fused["lm_heads.weight"] = torch.cat([sd[f"lm_heads.{k}.weight"] for k in range(spec.num_codebooks)], dim=0)
    enc = torch.randn(1, 5, spec.hidden_size, generator=torch.Generator().manual_seed(0))
    outs = []
    for w in (sd, fused):
        eng = make_engine(spec, w, torch.float32, max_batch=1)
        eng.set_gen_params(max_length=16)
        eng.prefill(enc, None, None, None, sample=False)
        outs.append(eng.logits().cpu())
    assert torch.equal(outs[0], outs[1])


def test_long_context_split_kv_matches_oracle():
    """Self-KV length grows past several 8-row batches per wave and several splits (no prompt, 150 steps)."""
    spec = DO.DecoderSpec(**{**DO.TINY.__dict__, "max_position_embeddings": 512})
    sd = DO.make_decoder_weights(spec, seed=21)
    g = torch.Generator().manual_seed(2)
    enc = torch.randn(1, 6, spec.hidden_size, generator=g)
    gp = DO.GenParams(max_length=160, min_new_tokens=159)
    ref = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, None, None, None, gp, keep_logits=True)
    eng = make_engine(spec, sd, torch.float32, max_batch=1, max_ctx=200)
    eng.set_gen_params(max_length=160, min_new_tokens=159)
    # teacher-forced on the oracle's own ids so a late tie cannot cascade: compare logits at every step
    eng.prefill(enc, None, None, None, sample=False)
    worst = float((eng.logits().cpu() - ref.step_logits[0]).abs().max())
    for s in range(1, 159):
        eng.push_tokens(ref.sequences[:, s])
        eng.step_forward()
        worst = max(worst, float((eng.logits().cpu() - ref.step_logits[s]).abs().max()))
    assert worst < 3e-5, worst


def test_sampling_topk1_equals_greedy_and_distribution():
    spec = DO.TINY
    sd = DO.make_decoder_weights(spec, seed=5)
    for k in range(spec.num_codebooks):
        sd[f"lm_heads.{k}.weight"] *= 12.0  # peaky distribution so a 3000-draw histogram is informative
    g = torch.Generator().manual_seed(9)
    enc = torch.randn(1, 5, spec.hidden_size, generator=g)
    eng = make_engine(spec, sd, torch.float32, max_batch=1)
    eng.set_gen_params(max_length=12, min_new_tokens=11)
    greedy = eng.generate_ids(enc, None, None, None).cpu()
    eng.set_gen_params(max_length=12, min_new_tokens=11, do_sample=True, top_k=1, seed=123)
    assert torch.equal(eng.generate_ids(enc, None, None, None).cpu(), greedy)
    # first-token distribution of codebook 0 under temperature 0.7 / top_k 20 / top_p 0.9 vs the restated warpers
    eng.prefill(enc, None, None, None, sample=False)
    logits0 = eng.logits().cpu()[0].clone()
    logits0[spec.eos_token_id] = -float("inf")  # min_new_tokens
    sc = logits0 / 0.7
    kth = torch.topk(sc, 20)[0][-1]
    sc = sc.masked_fill(sc < kth, -float("inf"))
    sl, si = torch.sort(sc, descending=False)
    rm = sl.softmax(-1).cumsum(-1) <= (1 - 0.9)
    rm[-1] = False
    sc = sc.masked_fill(rm.scatter(0, si, rm), -float("inf"))
    p = sc.softmax(-1)
    n = 3000
    counts = torch.zeros_like(p)
    for i in range(n):
        eng.set_gen_params(max_length=12, min_new_tokens=11, do_sample=True, temperature=0.7, top_k=20, top_p=0.9, seed=1000 + i)
        eng.prefill(enc, None, None, None, sample=True)
        counts[int(eng.ids()[0, 1])] += 1
    assert counts[p == 0].sum() == 0, "sampled a token outside the top-k/top-p support"
    sup = p > 0
    chi2 = (((counts[sup] - n * p[sup]) ** 2) / (n * p[sup])).sum()
    assert chi2 < 3 * int(sup.sum()) + 30, float(chi2)


def test_capacity_and_argument_errors_are_value_errors():
    spec = DO.TINY
    sd = DO.make_decoder_weights(spec, seed=1)
    eng = make_engine(spec, sd, torch.float32, max_batch=1, max_ctx=64, max_enc=8, max_prompt=4)
    enc = torch.randn(1, 5, spec.hidden_size)
    with pytest.raises(ValueError, match="before ptts_prefill"):
        eng.decode_steps(1)
    with pytest.raises(ValueError, match="max_ctx"):
        eng.set_gen_params(max_length=65)
    eng.set_gen_params(max_length=32)
    with pytest.raises(ValueError, match="max_batch"):
        eng.prefill(torch.randn(2, 5, spec.hidden_size), None, None, None)
    with pytest.raises(ValueError, match="max_enc"):
        eng.prefill(torch.randn(1, 9, spec.hidden_size), None, None, None)
    with pytest.raises(ValueError, match="prompt length"):
        eng.prefill(enc, None, torch.randn(1, 4, spec.hidden_size), None)
    with pytest.raises(ValueError, match="expected shape"):
        eng.load_weight("model.decoder.layers.0.fc1.weight", torch.zeros(3, 3))
    with pytest.raises(ValueError, match="unknown tensor"):
        eng.load_weight("model.decoder.layers.0.bogus.weight", torch.zeros(3, 3))
