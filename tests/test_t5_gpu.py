"""GPU: the HIP T5 description encoder (ptts_t5_*, csrc/ptts_t5.hip) through the C ABI against the oracle (oracle/t5_oracle.py, pinned against
the installed transformers T5EncoderModel) and against the committed transformers outputs (tests/golden/t5_tiny.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from helpers import log_parity
from oracle import t5_oracle as TO

pytestmark = pytest.mark.gpu
LOG = "r05_parity_t5.txt"


def make_t5(spec, sd, dtype=torch.float32, max_batch=4, max_len=64):
    from parler_tts_amd.engine import T5Engine

    e = T5Engine(vocab_size=spec.vocab_size, d_model=spec.d_model, d_kv=spec.d_kv, d_ff=spec.d_ff, num_layers=spec.num_layers, num_heads=spec.num_heads,
                 relative_attention_num_buckets=spec.relative_attention_num_buckets, relative_attention_max_distance=spec.relative_attention_max_distance,
                 layer_norm_epsilon=spec.layer_norm_epsilon, dtype=dtype, max_batch=max_batch, max_len=max_len)
    e.load_state_dict(sd)
    return e


def rel_rms(a, b):
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def test_golden_transformers_outputs_fp32():
    z = np.load(os.path.join(GOLD, "t5_tiny.npz"))
    spec = TO.T5Spec(**{k: (float(v) if k == "layer_norm_epsilon" else int(v)) for k, v in zip(z["spec_keys"].tolist(), z["spec_vals"].tolist())})
    sd = TO.make_t5_weights(spec, seed=int(z["seed"]))
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    eng = make_t5(spec, sd, max_batch=3, max_len=37)
    out = eng.encode(ids.cuda(), mask.cuda()).cpu()
    ref = torch.from_numpy(z["hf_masked"]) * mask[..., None].float()
    err = float((out - ref).abs().max())
    log_parity(f"[t5 golden fp32 masked] max |d| {err:.2e} vs transformers T5EncoderModel outputs", LOG)
    assert err <= 2e-5
    assert float(out[mask == 0].abs().max()) == 0.0  # masked positions are exactly zero
    out = eng.encode(ids.cuda(), None).cpu()
    err = float((out - torch.from_numpy(z["hf_unmasked"])).abs().max())
    log_parity(f"[t5 golden fp32 unmasked] max |d| {err:.2e}", LOG)
    assert err <= 2e-5


@pytest.mark.parametrize("B,N", [(1, 64), (1, 17), (3, 40), (5, 64), (2, 140)])
def test_mini_widths_fp32_vs_oracle(B, N):
    """flan-t5-large widths (d_model 1024, 16 heads, d_ff 2816), 2 blocks: one utterance on the strip kernels in fragment order (M <= 256),
    M = 320 on the block kernels (row-major); ragged masks incl. left padding; 140 tokens = three key tiles + saturated buckets."""
    spec = TO.T5Spec(vocab_size=512, d_model=1024, d_kv=64, d_ff=2816, num_layers=2, num_heads=16)
    sd = TO.make_t5_weights(spec, seed=7)
    g = torch.Generator().manual_seed(100 * B + N)
    ids = torch.randint(0, spec.vocab_size, (B, N), generator=g)
    mask = torch.ones(B, N, dtype=torch.long)
    if B > 1:
        mask[1, N - N // 3:] = 0
        mask[B - 1, : N // 4] = 0
    eng = make_t5(spec, sd, max_batch=B, max_len=N)
    ref = TO.T5Oracle(spec, sd).encode(ids, mask if B > 1 else None)
    out = eng.encode(ids.cuda(), mask.cuda() if B > 1 else None).cpu()
    err, r = float((out - ref).abs().max()), rel_rms(out, ref)
    log_parity(f"[t5 fp32 B={B} N={N}] max |d| {err:.2e}, relative RMS {r:.2e}", LOG)
    assert err <= 5e-5 and r <= 1e-5


@pytest.mark.parametrize("B,N", [(1, 64), (5, 64)])
def test_mini_widths_bf16_vs_bf16_oracle(B, N):
    """bf16 engine vs the oracle that rounds where the engine rounds; and the engine must be at least as close to the fp32 truth as the
    rounding model itself is (it keeps MORE in fp32 than transformers' bf16 run)."""
    spec = TO.T5Spec(vocab_size=512, d_model=1024, d_kv=64, d_ff=2816, num_layers=2, num_heads=16)
    sd = TO.make_t5_weights(spec, seed=7)
    g = torch.Generator().manual_seed(100 * B + N)
    ids = torch.randint(0, spec.vocab_size, (B, N), generator=g)
    mask = torch.ones(B, N, dtype=torch.long)
    if B > 1:
        mask[1, N - N // 3:] = 0
    eng = make_t5(spec, sd, dtype=torch.bfloat16, max_batch=B, max_len=N)
    out = eng.encode(ids.cuda(), mask.cuda()).cpu()
    ref16 = TO.T5Oracle(spec, sd, precision="bf16", fold_norm=B * N <= 256).encode(ids, mask)  # <= 256 rows: the engine folds the norms into the GEMMs
    ref32 = TO.T5Oracle(spec, sd).encode(ids, mask)
    r16, r32, model = rel_rms(out, ref16), rel_rms(out, ref32), rel_rms(ref16, ref32)
    log_parity(f"[t5 bf16 B={B} N={N}] relative RMS vs bf16 oracle {r16:.2e}, vs fp32 oracle {r32:.2e} (bf16 oracle vs fp32: {model:.2e})", LOG)
    assert r16 <= 3e-3
    assert r32 <= 1.5 * model + 1e-4


def test_full_depth_flan_t5_large_shape():
    """All 24 blocks at flan-t5-large widths, one 64-token description (bench.py's configuration), fp32 and bf16."""
    spec = TO.T5Spec(vocab_size=1024, d_model=1024, d_kv=64, d_ff=2816, num_layers=24, num_heads=16)
    sd = TO.make_t5_weights(spec, seed=9)
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(0, spec.vocab_size, (1, 64), generator=g)
    ref = TO.T5Oracle(spec, sd).encode(ids, None)
    out = make_t5(spec, sd, max_batch=1, max_len=64).encode(ids.cuda(), None).cpu()
    err, r = float((out - ref).abs().max()), rel_rms(out, ref)
    log_parity(f"[t5 fp32 24 blocks] max |d| {err:.2e}, relative RMS {r:.2e}", LOG)
    assert r <= 2e-5
    out = make_t5(spec, sd, dtype=torch.bfloat16, max_batch=1, max_len=64).encode(ids.cuda(), None).cpu()
    ref16 = TO.T5Oracle(spec, sd, precision="bf16", fold_norm=True).encode(ids, None)
    r16, r32, model = rel_rms(out, ref16), rel_rms(out, ref), rel_rms(ref16, ref)
    log_parity(f"[t5 bf16 24 blocks] relative RMS vs bf16 oracle {r16:.2e}, vs fp32 {r32:.2e} (bf16 oracle vs fp32 {model:.2e})", LOG)
    assert r16 <= 1e-2 and r32 <= 1.5 * model + 1e-4


def test_graph_replay_equals_eager_and_repeats():
    """The captured graph (default) and the eager launch list (PTTS_T5_NO_GRAPH) produce identical bits; a second call with other ids on the
    same shape replays the same graph."""
    spec = TO.T5Spec(vocab_size=200)
    sd = TO.make_t5_weights(spec, seed=1)
    g = torch.Generator().manual_seed(2)
    ids1 = torch.randint(0, 200, (2, 33), generator=g).cuda()
    ids2 = torch.randint(0, 200, (2, 33), generator=g).cuda()
    eng = make_t5(spec, sd, max_batch=2, max_len=33)
    a1, a2, a1b = eng.encode(ids1).clone(), eng.encode(ids2).clone(), eng.encode(ids1).clone()
    assert torch.equal(a1, a1b) and not torch.equal(a1, a2)
    os.environ["PTTS_T5_NO_GRAPH"] = "1"
    try:
        eager = make_t5(spec, sd, max_batch=2, max_len=33)
    finally:
        del os.environ["PTTS_T5_NO_GRAPH"]
    assert torch.equal(eager.encode(ids1), a1)
    ref = TO.T5Oracle(spec, sd).encode(ids1.cpu(), None)
    assert float((a1.cpu() - ref).abs().max()) <= 2e-5


@pytest.mark.parametrize("dtype,prec", [(torch.float32, "fp32"), (torch.bfloat16, "bf16")])
def test_norm_folded_into_the_gemms_vs_rows_prep_nodes(dtype, prec):
    """<= 256 rows: T5LayerNorm folded into the GEMMs around it (the o / wo residual epilogues emit g o h and per-strip sums of squares, the q|k|v and
    wi GEMMs scale their accumulators by rstd; default) against the rows_prep node in front of every GEMM (PTTS_T5_NO_FOLD=1), each against the oracle
    that rounds where it rounds; 3 blocks (block 0's first norm is never folded), ragged masks, 3 x 40 = 120 rows."""
    spec = TO.T5Spec(vocab_size=300, d_model=1024, d_kv=64, d_ff=2816, num_layers=3, num_heads=16)
    sd = TO.make_t5_weights(spec, seed=13)
    g = torch.Generator().manual_seed(14)
    ids = torch.randint(0, 300, (3, 40), generator=g)
    mask = torch.ones(3, 40, dtype=torch.long)
    mask[0, 33:] = 0
    mask[2, :7] = 0
    outs = {}
    for fold in (True, False):
        os.environ["PTTS_T5_NO_FOLD"] = "0" if fold else "1"
        try:
            eng = make_t5(spec, sd, dtype=dtype, max_batch=3, max_len=40)
        finally:
            del os.environ["PTTS_T5_NO_FOLD"]
        outs[fold] = eng.encode(ids.cuda(), mask.cuda()).cpu()
        ref = TO.T5Oracle(spec, sd, precision=prec, fold_norm=fold).encode(ids, mask)
        r = rel_rms(outs[fold], ref)
        log_parity(f"[t5 {prec} norm {'folded' if fold else 'rows_prep'}] relative RMS vs its oracle {r:.2e}", LOG)
        assert r <= (1e-5 if prec == "fp32" else 3e-3), (fold, r)
    ab = rel_rms(outs[True], outs[False])
    assert 0.0 < ab <= (1e-5 if prec == "fp32" else 1e-2), ab  # the folded nodes really ran, and re-associate / re-round only


def test_capacity_and_name_errors():
    spec = TO.T5Spec(vocab_size=200)
    sd = TO.make_t5_weights(spec, seed=1)
    eng = make_t5(spec, sd, max_batch=1, max_len=16)
    with pytest.raises(ValueError):
        eng.encode(torch.zeros(2, 8, dtype=torch.long).cuda())
    with pytest.raises(ValueError):
        eng.encode(torch.zeros(1, 17, dtype=torch.long).cuda())
    from parler_tts_amd.engine import T5Engine

    e2 = T5Engine(vocab_size=200, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2, dtype=torch.float32, max_batch=1, max_len=16)
    with pytest.raises(ValueError, match="not loaded"):
        e2.encode(torch.zeros(1, 8, dtype=torch.long).cuda())
    with pytest.raises(NotImplementedError):
        T5Engine(vocab_size=200, d_model=128, d_kv=32, d_ff=256, num_layers=2, num_heads=4, dtype=torch.float32)


def test_generate_uses_native_encoder_and_matches_stock_module():
    """The model-level switch: the description encoding generate() feeds to the decoder comes from the HIP encoder by default and equals the
    stock transformers module's (fp32) on the same weights; masked positions zero on both paths."""
    import parler_tts_amd as P
    from transformers import T5Config

    t5 = T5Config(vocab_size=300, d_model=128, d_kv=64, d_ff=256, num_layers=2, num_heads=2, feed_forward_proj="gated-gelu", tie_word_embeddings=False)
    dec = P.ParlerTTSDecoderConfig(vocab_size=1088, max_position_embeddings=512, num_hidden_layers=2, ffn_dim=256, num_attention_heads=2, hidden_size=128,
                                   num_codebooks=9, pad_token_id=1024, eos_token_id=1024, bos_token_id=1025)
    torch.manual_seed(3)
    model = P.ParlerTTSForConditionalGeneration(P.ParlerTTSConfig.from_sub_models_config(t5, P.DACConfig(), dec, vocab_size=300)).to("cuda")
    ids = torch.randint(3, 300, (2, 19)).cuda()
    mask = torch.ones(2, 19, dtype=torch.long).cuda()
    mask[1, 12:] = 0
    native = model._encode_description(ids, mask).float()
    assert model.__dict__.get("_t5_engine") is not None
    model.use_native_text_encoder = False
    stock = model._encode_description(ids, mask).float()
    err = float((native - stock).abs().max())
    log_parity(f"[t5 in generate()] native vs stock module, fp32: max |d| {err:.2e}", LOG)
    assert err <= 5e-5
    assert float(native[mask == 0].abs().max()) == 0.0


def test_bench_ttft_shape_32_descriptions_24_blocks_bf16():
    """the `ttft` object of the bench line at batch 32 at ITS OWN shape - flan-t5-large widths, all 24 blocks, 32 descriptions x
    64 tokens = 2048 rows (the > 256-row GEMM path and the tiled attention), bf16 engine against the bf16-rounding oracle and against the fp32
    oracle; ragged masks on two rows (right- and left-padded)."""
    spec = TO.T5Spec(vocab_size=1024, d_model=1024, d_kv=64, d_ff=2816, num_layers=24, num_heads=16)
    sd = TO.make_t5_weights(spec, seed=9)
    g = torch.Generator().manual_seed(3264)
    B, N = 32, 64
    ids = torch.randint(0, spec.vocab_size, (B, N), generator=g)
    mask = torch.ones(B, N, dtype=torch.long)
    mask[1, 41:] = 0
    mask[31, :9] = 0
    out = make_t5(spec, sd, dtype=torch.bfloat16, max_batch=B, max_len=N).encode(ids.cuda(), mask.cuda()).cpu()
    ref16 = TO.T5Oracle(spec, sd, precision="bf16", fold_norm=False).encode(ids, mask)  # > 256 rows: rows_prep nodes, no folded norms
    ref32 = TO.T5Oracle(spec, sd).encode(ids, mask)
    r16, r32, model = rel_rms(out, ref16), rel_rms(out, ref32), rel_rms(ref16, ref32)
    err16 = float((out - ref16).abs().max())
    log_parity(f"[t5 bf16 24 blocks, 32 x 64 tokens = 2048 rows] relative RMS vs bf16 oracle {r16:.2e} (max |d| {err16:.2e}), vs fp32 {r32:.2e} "
               f"(bf16 oracle vs fp32 {model:.2e})", "r06_parity_t5.txt")
    assert float(out[mask == 0].abs().max()) == 0.0
    assert r16 <= 1e-2 and r32 <= 1.5 * model + 1e-4
    out32 = make_t5(spec, sd, max_batch=B, max_len=N).encode(ids.cuda(), mask.cuda()).cpu()
    e32, q32 = float((out32 - ref32).abs().max()), rel_rms(out32, ref32)
    log_parity(f"[t5 fp32 24 blocks, 32 x 64 tokens] max |d| {e32:.2e}, relative RMS {q32:.2e}", "r06_parity_t5.txt")
    assert q32 <= 2e-5


@pytest.mark.parametrize("B,N", [(2, 64), (3, 40), (2, 140), (1, 17)])
def test_f32_mfma_attention_kernel_vs_oracle_and_vs_valu_kernel(B, N, monkeypatch):
    """t5_attn_mfma_kernel (round 6: exact-f32 MFMA, 64 queries per workgroup; the default from 128 workgroups up) forced on small shapes
    (PTTS_T5_ATTN_MFMA=1) against the oracle and against the VALU kernel (=0; same arithmetic, another summation order): 64 tokens = one full key
    block, 40 / 17 = a partial block (absent keys, clamped queries), 140 = three blocks (online softmax across blocks, saturated buckets); ragged
    masks with left and right padding and one FULLY masked description (uniform attention, as the additive mask of the reference leaves it); fp32
    engine (bit-level arithmetic visible) and bf16 engine (bf16 context rows written by the kernel). Graphs off: the switch is read per forward."""
    monkeypatch.setenv("PTTS_T5_NO_GRAPH", "1")
    spec = TO.T5Spec(vocab_size=300, d_model=1024, d_kv=64, d_ff=2816, num_layers=2, num_heads=16)
    sd = TO.make_t5_weights(spec, seed=21)
    g = torch.Generator().manual_seed(1000 * B + N)
    ids = torch.randint(0, 300, (B, N), generator=g)
    mask = torch.ones(B, N, dtype=torch.long)
    if B > 1:
        mask[1, N - N // 3:] = 0
        mask[0, : N // 5] = 0
    if B > 2:
        mask[2, :] = 0  # fully masked row
    ref = TO.T5Oracle(spec, sd).encode(ids, mask)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PTTS_T5_ATTN_MFMA", mode)
        eng = make_t5(spec, sd, max_batch=B, max_len=N)
        outs[mode] = eng.encode(ids.cuda(), mask.cuda()).cpu()
    err, ab = float((outs["1"] - ref).abs().max()), float((outs["1"] - outs["0"]).abs().max())
    log_parity(f"[t5 f32-MFMA attention fp32 B={B} N={N}] max |d| vs oracle {err:.2e}, vs the VALU kernel {ab:.2e}", "r06_parity_t5.txt")
    assert err <= 5e-5 and ab <= 5e-5
    monkeypatch.setenv("PTTS_T5_ATTN_MFMA", "1")
    out16 = make_t5(spec, sd, dtype=torch.bfloat16, max_batch=B, max_len=N).encode(ids.cuda(), mask.cuda()).cpu()
    ref16 = TO.T5Oracle(spec, sd, precision="bf16", fold_norm=B * N <= 256).encode(ids, mask)
    r16 = rel_rms(out16, ref16)
    log_parity(f"[t5 f32-MFMA attention bf16 B={B} N={N}] relative RMS vs bf16 oracle {r16:.2e}", "r06_parity_t5.txt")
    assert r16 <= 3e-3
