"""CPU: product-side host logic (delay pattern, EOS gate, configs, checkpoint names) against the oracle and the
golden vectors — no GPU, no HIP compute."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLD
from helpers import t
from oracle import decoder_oracle as DO

import parler_tts_amd as P


def test_delay_pattern_matches_reference_known_answers():
    g = np.load(os.path.join(GOLD, "delay_kat.npz"))
    for ci in range(int(g["n"])):
        K, seq_len, max_len, bsz = [int(x) for x in g[f"c{ci}_args"]]
        ids, mask = P.build_delay_pattern_mask(t(g[f"c{ci}_in"]), 1025, 1024, max_len, K)
        assert torch.equal(ids, t(g[f"c{ci}_ids"])), ci
        assert torch.equal(mask, t(g[f"c{ci}_mask"])), ci


def test_delay_pattern_random_differential_vs_oracle():
    gen = torch.Generator().manual_seed(0)
    for _ in range(60):
        K = int(torch.randint(1, 10, (1,), generator=gen))
        bsz = int(torch.randint(1, 4, (1,), generator=gen))
        max_len = int(torch.randint(1, 40, (1,), generator=gen))
        seq_len = int(torch.randint(1, max(2, min(max_len, 6)), (1,), generator=gen))
        ids = torch.randint(0, 1024, (bsz * K, seq_len), generator=gen)
        a = P.build_delay_pattern_mask(ids, 1025, 1024, max_len, K)
        b = DO.build_delay_pattern_mask(ids, 1025, 1024, max_len, K)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (K, bsz, max_len, seq_len)
        full = torch.randint(0, 1024, (bsz * K, max_len), generator=gen)
        assert torch.equal(P.apply_delay_pattern_mask(full, a[1]), DO.apply_delay_pattern_mask(full, b[1]))


def test_logits_processor_matches_reference_known_answers():
    g = np.load(os.path.join(GOLD, "eosgate_kat.npz"))
    K, bsz = int(g["K"]), int(g["bsz"])
    proc = P.ParlerTTSLogitsProcessor(1024, K, bsz, "cpu")
    hist = t(g["history"])
    for s in range(g["gated"].shape[0]):
        sc = proc(hist[:, : s + 2], torch.zeros(bsz * K, 1088))
        assert np.array_equal(torch.isinf(sc[:, 1024]).numpy(), g["gated"][s])
        assert torch.isinf(sc).sum() == int(g["gated"][s].sum())  # only the EOS column is touched
    with pytest.raises(ValueError, match="positive integers"):
        P.ParlerTTSLogitsProcessor([-1], K, bsz)


def _tiny_config(**dec_kw):
    from transformers import T5Config

    t5 = T5Config(vocab_size=128, d_model=128, d_kv=32, d_ff=256, num_layers=2, num_heads=4, feed_forward_proj="gated-gelu")
    dec = P.ParlerTTSDecoderConfig(vocab_size=1088, max_position_embeddings=256, num_hidden_layers=2, ffn_dim=256, num_attention_heads=2,
                                   hidden_size=128, num_codebooks=9, pad_token_id=1024, eos_token_id=1024, bos_token_id=1025, **dec_kw)
    dac = P.DACConfig(latent_dim=64, decoder_dim=256, decoder_rates=[4, 2, 2, 2])
    return P.ParlerTTSConfig.from_sub_models_config(t5, dac, dec, vocab_size=128)


def test_config_round_trip_and_reference_field_names(tmp_path):
    cfg = _tiny_config()
    cfg.save_pretrained(str(tmp_path))
    raw = json.load(open(tmp_path / "config.json"))
    assert raw["model_type"] == "parler_tts" and raw["decoder"]["model_type"] == "parler_tts_decoder"
    assert raw["audio_encoder"]["model_type"] == "dac_on_the_hub" and raw["text_encoder"]["model_type"] == "t5"
    for f in ("vocab_size", "max_position_embeddings", "num_hidden_layers", "ffn_dim", "num_attention_heads", "num_key_value_heads",
              "hidden_size", "num_codebooks", "rope_embeddings", "rope_theta", "use_fused_lm_heads", "pad_token_id", "bos_token_id", "eos_token_id"):
        assert f in raw["decoder"], f
    back = P.ParlerTTSConfig.from_pretrained(str(tmp_path))
    assert back.decoder.to_dict() == cfg.decoder.to_dict()
    assert back.sampling_rate == 44100 and back.audio_encoder.frame_rate == 86
    with pytest.raises(ValueError, match="text_encoder, audio_encoder and decoder"):
        P.ParlerTTSConfig(vocab_size=10)
    with pytest.raises(ValueError, match="codebook_weights"):
        P.ParlerTTSDecoderConfig(num_codebooks=4, codebook_weights=[1.0])


def test_state_dict_uses_reference_names_and_round_trips(tmp_path):
    torch.manual_seed(0)
    m = P.ParlerTTSForConditionalGeneration(_tiny_config())
    from oracle import dac_oracle as DA

    m.audio_encoder.load_state_dict({"model." + k: v for k, v in DA.make_dac_weights(DA.DAC_TINY, 1, "parametrized").items()})
    keys = set(m.state_dict().keys())
    for k in ("decoder.model.decoder.layers.1.self_attn.q_proj.weight", "decoder.model.decoder.layers.0.encoder_attn_layer_norm.bias",
              "decoder.model.decoder.embed_tokens.8.weight", "decoder.model.decoder.embed_positions.weights", "decoder.lm_heads.3.weight",
              "decoder.model.decoder.layer_norm.weight", "embed_prompts.weight", "decoder.model.decoder.layers.1.fc2.weight"):
        assert k in keys, k
    assert any(k.startswith("text_encoder.") for k in keys)
    m.save_pretrained(str(tmp_path))
    m2 = P.ParlerTTSForConditionalGeneration.from_pretrained(str(tmp_path))
    a, b = m.state_dict(), m2.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a if not k.endswith("_dummy"))
    assert "model.decoder.model.1.block.1.parametrizations.weight.original0" in m2.audio_encoder.state_dict()
    assert m2.generation_config.decoder_start_token_id == 1025 and m2.generation_config.max_length == 2580
    # loader kwargs: `dtype=` alias, single-device `device_map`, config-field overrides; anything else is an error, not silently dropped
    m3 = P.ParlerTTSForConditionalGeneration.from_pretrained(str(tmp_path), dtype="bfloat16", device_map={"": "cpu"}, low_cpu_mem_usage=True,
                                                             attn_implementation="sdpa", vocab_size=m.config.vocab_size)
    assert m3.dtype == torch.bfloat16 and m3.device.type == "cpu"
    assert torch.equal(m3.state_dict()["embed_prompts.weight"], a["embed_prompts.weight"].to(torch.bfloat16))
    with pytest.raises(TypeError, match="unexpected keyword argument 'torch_dtyp'"):
        P.ParlerTTSForConditionalGeneration.from_pretrained(str(tmp_path), torch_dtyp=torch.bfloat16)
    with pytest.raises(NotImplementedError, match="one device"):
        P.ParlerTTSForConditionalGeneration.from_pretrained(str(tmp_path), device_map={"text_encoder": 0, "decoder": 1})
    with pytest.raises(NotImplementedError, match="float16"):
        P.ParlerTTSForConditionalGeneration.from_pretrained(str(tmp_path), torch_dtype=torch.float16)


def test_unsupported_architectures_fail_loudly():
    with pytest.raises(ValueError, match="divisible"):
        P.ParlerTTSForConditionalGeneration(_tiny_config(num_key_value_heads=3))
    m = P.ParlerTTSForConditionalGeneration(_tiny_config(num_key_value_heads=1))  # grouped-query attention is supported: K/V rows shrink
    sd = m.decoder.state_dict()
    assert sd["model.decoder.layers.0.self_attn.k_proj.weight"].shape[0] == 64 and sd["model.decoder.layers.0.self_attn.q_proj.weight"].shape[0] == 128
    with pytest.raises(NotImplementedError, match="gelu"):
        P.ParlerTTSForConditionalGeneration(_tiny_config(activation_function="relu"))


def test_generate_refuses_cpu_and_bad_modes():
    m = P.ParlerTTSForConditionalGeneration(_tiny_config())
    ids = torch.randint(3, 100, (1, 6))
    with pytest.raises(ValueError, match="greedy or sampling"):
        m.generate(input_ids=ids, prompt_input_ids=ids, num_beams=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):  # voice prompt: DAC encode is HIP-only too
        m.generate(input_ids=ids, prompt_input_ids=ids, input_values=torch.zeros(1, 1, 100))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.generate(input_ids=ids, prompt_input_ids=ids, max_new_tokens=12)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.audio_encoder.encode(torch.zeros(1, 1, 100))
    with pytest.raises(NotImplementedError, match="resize"):
        m.resize_token_embeddings(10)


def test_voice_prompt_delay_round_trip_matches_reference_undelay_mask():
    """With a voice prompt the delay pattern holds the shifted prompt (modeling:205-276); un-delaying with the BOS/PAD
    triangles returns it intact. The reference rebuilds its un-delay mask from the UN-delayed `input_ids` (BOS column +
    prompt codes, :3589-3594) and keeps positions that are neither BOS nor PAD (:3596): audio codes are never BOS/PAD,
    so that keep-mask is exactly the one of the BOS column alone, which is what generate() here builds."""
    K, L, T, bos, pad = 9, 40, 12, 1025, 1024
    codes = (torch.arange(K * T).reshape(K, T) * 7) % 1000
    dec = torch.cat([torch.full((K, 1), bos), codes], 1)
    delayed, pattern = P.build_delay_pattern_mask(dec, bos, pad, L, K)
    assert delayed.shape == (K, T + 1)
    raw = torch.full((K, L), 7)
    raw[:, : T + 1] = delayed
    out = P.apply_delay_pattern_mask(raw, pattern)
    _, m_ok = P.build_delay_pattern_mask(dec[:, :1], bos, pad, L, K)
    keep_ok = (m_ok != bos) & (m_ok != pad)
    und = out[keep_ok].reshape(1, K, -1)
    assert und.shape[-1] == L - K and torch.equal(und[0, :, :T], codes)
    _, m_ref = P.build_delay_pattern_mask(dec, bos, pad, L, K)  # what the reference passes (:3589): the un-delayed ids
    keep_ref = (m_ref != bos) & (m_ref != pad)
    assert torch.equal(keep_ref, keep_ok)


class _FakeCodec:
    """CPU stand-in for DACModel.decode with a finite, symmetric receptive field (3 latent frames each side, like a short
    conv stack): sample n of frame t depends on frames t-3..t+3 and on every codebook. Lets the streamer's host logic
    (un-delay, special-id filter, halo windows, stride trimming) be tested without a GPU."""

    hop = 32
    decoder_rates = (4, 2, 2, 2)

    class config:
        sampling_rate, frame_rate, codebook_size, num_codebooks = 32 * 86, 86, 1024, 9

    device = torch.device("cpu")

    def decode(self, audio_codes, audio_scales=None):
        codes = audio_codes[0].double()  # [B, K, T]
        B, K, T = codes.shape
        w = torch.linspace(0.3, 1.0, K, dtype=torch.float64)[None, :, None]
        lat = (codes * w).sum(1) / 1024.0  # [B, T]
        pad = torch.nn.functional.pad(lat, (3, 3))
        taps = torch.tensor([0.05, -0.1, 0.3, 1.0, 0.25, -0.15, 0.07], dtype=torch.float64)
        sm = sum(taps[i] * pad[:, i: i + T] for i in range(7))  # [B, T]
        phase = torch.arange(self.hop, dtype=torch.float64) / self.hop
        wav = (sm[:, :, None] * torch.cos(2 * torch.pi * phase)[None, None] + 0.1 * pad[:, 2: 2 + T, None] * phase[None, None]).reshape(B, 1, T * self.hop)

        class Out:
            audio_values = wav.float()

        return Out()


def test_streamer_incremental_equals_full_redecode_on_cpu():
    """ParlerTTSStreamer (streamer.py:66-131): the chunks emitted with the O(n) halo-window decode are the chunks of the
    reference's full re-decode every `play_steps`, incl. the stride trimming, a frame dropped for a special id and the tail."""
    import types
    from parler_tts_amd.streamer import ParlerTTSStreamer, receptive_halo_frames

    assert receptive_halo_frames((8, 8, 4, 2)) == 13  # documented halo of the 44 kHz decoder
    K = 9
    gc = types.SimpleNamespace(bos_token_id=1025, pad_token_id=1024, decoder_start_token_id=1025)
    model = types.SimpleNamespace(decoder=types.SimpleNamespace(num_codebooks=K), audio_encoder=_FakeCodec(), generation_config=gc,
                                  device=torch.device("cpu"), use_audio_scales=False, use_4dim_audio_codes=True)
    g = torch.Generator().manual_seed(0)
    L = 70
    raw = torch.randint(0, 1024, (K, L), generator=g)
    raw[3, 40] = 1030  # one special id mid-stream: that frame is dropped by both variants
    chunks = {}
    for inc in (False, True):
        s = ParlerTTSStreamer(model, play_steps=10, incremental=inc)
        s.halo_frames = 3 + 1 if inc else s.halo_frames  # the fake codec's receptive field
        first = P.build_delay_pattern_mask(torch.full((K, 1), 1025), 1025, 1025, L, K)[0]
        s.put(first)
        for j in range(1, L):
            s.put(raw[:, j])
        s.end()
        out = []
        for c in s:
            out.append(c)
        chunks[inc] = out
    assert len(chunks[True]) == len(chunks[False]) and len(chunks[True]) >= 7
    for a, b in zip(chunks[True], chunks[False]):
        assert a.shape == b.shape and np.allclose(a, b, atol=1e-6), (a.shape, b.shape)
    total = np.concatenate(chunks[True])
    assert total.shape[0] > 0 and total.shape[0] % 32 == 0
    with pytest.raises(ValueError, match="batch size 1"):
        ParlerTTSStreamer(model, play_steps=10).put(torch.zeros(2 * K, dtype=torch.long))


def test_host_side_tables_and_weight_norm_match_the_oracle():
    from oracle import dac_oracle as DA
    from parler_tts_amd.engine import fold_weight_norm, rope_tables
    from parler_tts_amd.synthetic import random_dac_state_dict

    c1, s1 = rope_tables(64, 10000.0, 300)
    c2, s2 = DO.rope_tables(64, 10000.0, 300)
    assert torch.equal(c1, c2) and torch.equal(s1, s2)
    for fmt in ("legacy", "parametrized"):
        sd = DA.make_dac_weights(DA.DAC_TINY, 7, fmt, with_encoder=True)
        a, b = fold_weight_norm(sd), DA.fold_weight_norm(sd)
        assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    sd = random_dac_state_dict(latent_dim=64, decoder_dim=256, rates=(4, 2, 2, 2))
    assert set(fold_weight_norm(sd)) == set(DA.fold_weight_norm(DA.make_dac_weights(DA.DAC_TINY, 1)))  # same tensor names as the oracle's decoder


def test_from_sub_models_pretrained_round_trip(tmp_path):
    """helpers/model_init_scripts/init_model_600M.py workflow: decoder.save_pretrained -> from_sub_models_pretrained(text encoder dir,
    audio encoder dir, decoder dir, vocab_size=...) -> save_pretrained -> from_pretrained; same tensors under the reference's names."""
    from transformers import T5Config, T5EncoderModel

    from oracle import dac_oracle as DA

    t5 = T5EncoderModel(T5Config(vocab_size=128, d_model=128, d_kv=32, d_ff=256, num_layers=2, num_heads=4, feed_forward_proj="gated-gelu")).eval()
    t5.save_pretrained(str(tmp_path / "t5"))
    dac = P.DACModel(P.DACConfig(latent_dim=64, decoder_dim=256, decoder_rates=[4, 2, 2, 2], encoder_dim=16))
    dac.load_state_dict({"model." + k: v for k, v in DA.make_dac_weights(DA.DAC_TINY, 4321, "parametrized", with_encoder=True).items()})
    dac.save_pretrained(str(tmp_path / "dac"))
    dec_cfg = P.ParlerTTSDecoderConfig(vocab_size=1088, max_position_embeddings=256, num_hidden_layers=2, ffn_dim=256, num_attention_heads=2,
                                       hidden_size=128, num_codebooks=9, pad_token_id=1024, eos_token_id=1024, bos_token_id=1025)
    dec = P.ParlerTTSForCausalLM(dec_cfg)
    dec.save_pretrained(str(tmp_path / "decoder"))
    m = P.ParlerTTSForConditionalGeneration.from_sub_models_pretrained(
        text_encoder_pretrained_model_name_or_path=str(tmp_path / "t5"), audio_encoder_pretrained_model_name_or_path=str(tmp_path / "dac"),
        decoder_pretrained_model_name_or_path=str(tmp_path / "decoder"), vocab_size=128)
    assert m.config.vocab_size == 128 and m.config.decoder.num_codebooks == 9 and m.config.audio_encoder.encoder_dim == 16
    for k, v in dec.state_dict().items():
        assert torch.equal(m.decoder.state_dict()[k], v), k
    assert torch.equal(m.audio_encoder.state_dict()["model.decoder.model.0.bias"], dac.state_dict()["model.decoder.model.0.bias"])
    assert m.get_decoder() is m.decoder and m.get_encoder() is m.text_encoder and m.get_input_embeddings() is m.text_encoder.get_input_embeddings()
    m.save_pretrained(str(tmp_path / "full"))
    m2 = P.ParlerTTSForConditionalGeneration.from_pretrained(str(tmp_path / "full"))
    a, b = m.state_dict(), m2.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a if not k.endswith("_dummy"))
    import shutil

    legacy = tmp_path / "legacy"  # torch.save checkpoint of the same tensors
    legacy.mkdir()
    shutil.copy(tmp_path / "full" / "config.json", legacy / "config.json")
    from safetensors.torch import load_file

    torch.save(load_file(str(tmp_path / "full" / "model.safetensors")), str(legacy / "pytorch_model.bin"))
    m3 = P.ParlerTTSForConditionalGeneration.from_pretrained(str(legacy))
    assert all(torch.equal(a[k], m3.state_dict()[k]) for k in a if not k.endswith("_dummy"))
    (tmp_path / "empty").mkdir()
    shutil.copy(tmp_path / "full" / "config.json", tmp_path / "empty" / "config.json")
    with pytest.raises(FileNotFoundError, match="pytorch_model.bin"):
        P.ParlerTTSForConditionalGeneration.from_pretrained(str(tmp_path / "empty"))
    d2 = P.ParlerTTSForCausalLM.from_pretrained(str(tmp_path / "full"))  # a full checkpoint also serves as a decoder source (:2654-2666)
    assert torch.equal(d2.state_dict()["lm_heads.3.weight"], dec.state_dict()["lm_heads.3.weight"])
    with pytest.raises(ValueError, match="decoder_pretrained_model_name_or_path"):
        P.ParlerTTSForConditionalGeneration.from_sub_models_pretrained(text_encoder_model=t5, audio_encoder_model=dac)
    labels = torch.tensor([[[5, 6], [7, -100], [9, 10]]])  # [bsz, seq, codebooks]
    assert m.prepare_decoder_input_ids_from_labels(labels).tolist() == [[[1025, 5, 7], [1025, 6, 1024]]]
    with pytest.raises(NotImplementedError, match="generate"):
        m(input_ids=torch.zeros(1, 2, dtype=torch.long))


def test_dac_codec_is_registered_with_transformers_auto_classes(tmp_path):
    """The reference's codec plug point (parler_tts/__init__.py:20-25): AutoConfig.register("dac_on_the_hub", DACConfig) and
    AutoModel.register(DACConfig, DACModel): a DAC directory resolves to this package's config / codec through the Auto classes."""
    from oracle import dac_oracle as DA
    from transformers import AutoConfig, AutoModel

    assert P.REGISTERED_WITH_TRANSFORMERS
    cfg = P.DACConfig(latent_dim=64, decoder_dim=256, decoder_rates=[4, 2, 2, 2], encoder_dim=16)
    m = P.DACModel(cfg)
    m.load_state_dict({"model." + k: v for k, v in DA.make_dac_weights(DA.DAC_TINY, seed=1).items()})
    m.save_pretrained(str(tmp_path))
    c2 = AutoConfig.from_pretrained(str(tmp_path))
    assert c2.model_type == "dac_on_the_hub" and c2.latent_dim == 64 and c2.codebook_size == 1024 and c2.frame_rate == 86
    m2 = AutoModel.from_pretrained(str(tmp_path))
    assert isinstance(m2, P.DACModel) and m2.decoder_dim == 256 and tuple(m2.decoder_rates) == (4, 2, 2, 2)
    assert set(m2.state_dict()) == set(m.state_dict())


def test_decoder_engines_are_cached_per_batch_class_and_only_grow(monkeypatch):
    """generate() keeps one HIP decoder engine per batch-size class (<= 4 utterances: GEMV step, 4 KV splits; wider: MFMA strips, no
    split) with grow-only capacities: alternating a wide batch with a long single utterance re-creates nothing, a single-utterance call
    never lands on an engine tuned for 32, and `model._engine = None` (weights / device / dtype changed) drops them all."""
    import types

    from parler_tts_amd import modeling_parler_tts as M

    made = []

    class FakeEngine:
        def __init__(self, **kw):
            self.cfg = types.SimpleNamespace(**{k: kw[k] for k in ("max_batch", "max_ctx", "max_enc", "max_prompt")})
            self.closed = False
            made.append(self)

        def load_state_dict(self, sd):
            self.loaded = len(sd)

        def close(self):
            self.closed = True

    monkeypatch.setattr(M, "DecoderEngine", FakeEngine)
    m = P.ParlerTTSForConditionalGeneration(_tiny_config())
    monkeypatch.setattr(type(m), "device", property(lambda self: torch.device("cuda", 0)))
    one = m._get_engine(1, 64, 32, 869)
    wide = m._get_engine(32, 64, 32, 100)
    assert m._get_engine(1, 64, 32, 869) is one and m._get_engine(20, 10, 5, 64) is wide and len(made) == 2  # alternating: nothing re-created
    assert (one.cfg.max_batch, wide.cfg.max_batch) == (1, 32) and m._engine is wide                        # last used
    grown = m._get_engine(2, 64, 40, 1200)  # same class, more capacity: re-created once with the maximum of old and new
    assert one.closed and len(made) == 3 and vars(grown.cfg) == dict(max_batch=2, max_ctx=1240, max_enc=64, max_prompt=41)
    assert m._get_engine(1, 16, 8, 64) is grown and m._get_engine(4, 64, 40, 1200) is not grown and len(made) == 4
    mid = m._get_engine(8, 16, 8, 64)  # 5..8 utterances: a class of its own (2 KV splits), neither the GEMV-step engine nor the wide one
    assert mid is not wide and mid.cfg.max_batch == 8 and m._get_engine(6, 16, 8, 64) is mid and m._get_engine(9, 16, 8, 64) is wide
    made.remove(mid)
    m._engine = None
    assert m._engine is None and m._get_engine(1, 16, 8, 64) not in made[:4] and len(made) == 5
    m.enable_fp8_weights(False)  # any weight-format / placement change drops the cache as well
    assert m._engine is None


def test_dac_engine_capacities_grow_within_a_memory_bound(monkeypatch):
    from parler_tts_amd.dac_wrapper import modeling_dac as MD

    made = []

    class FakeDac:
        def __init__(self, **kw):
            self.max_batch, self.max_frames, self.encoder_dim, self.compute_dtype = kw["max_batch"], kw["max_frames"], kw["encoder_dim"], kw["compute_dtype"]
            made.append(self)

        def load_state_dict(self, sd):
            pass

        def close(self):
            self.closed = True

    monkeypatch.setattr(MD, "DacEngine", FakeDac)
    from oracle import dac_oracle as DA

    d = P.DACModel(P.DACConfig(latent_dim=64), decoder_dim=256, decoder_rates=[4, 2, 2, 2])
    d.load_state_dict({"model." + k: v for k, v in DA.make_dac_weights(DA.DAC_TINY, 1).items()})
    monkeypatch.setattr(type(d), "device", property(lambda self: torch.device("cuda", 0)))
    a = d._get_engine(1, 2580)
    b = d._get_engine(8, 860)       # 8 x 2580 = 20640 frame-utterances: inside the bound, the engine keeps room for both shapes
    assert (b.max_batch, b.max_frames) == (8, 2580) and d._get_engine(1, 2580) is b and d._get_engine(8, 860) is b and len(made) == 2
    c = d._get_engine(32, 860)      # 32 x 2580 would be ~65 GB of activations: sized exactly instead
    assert (c.max_batch, c.max_frames) == (32, 860) and len(made) == 3 and a is not b
    big = d._get_engine(32, 2580)   # a request beyond the bound is served in sub-batches (decode() loops over the engine's max_batch)
    assert (big.max_batch, big.max_frames) == (11, 2580) and big.max_batch * big.max_frames <= P.DACModel.MAX_GROWN_FRAME_UTTERANCES
    assert d._get_engine(128, 2580) is big and len(made) == 4
    whole = d._get_engine(32, 100, whole_batch=True)  # chunked decode writes every utterance of the chunk in one pass
    assert whole.max_batch == 32 and whole.max_frames >= 100


def test_library_reads_fifteen_switches():
    """the product library reads at most 15 PTTS_* environment variables (56 in round 5), every one of them is a row of the table in
    DESIGN.md section 6 (with the test that covers both sides) and is set by a test under tests/; everything else goes through ptts_dev_env(), which is
    compiled out of the product build (-DPTTS_DEV_KNOBS builds only), and the built library carries no development-knob name at all."""
    import glob
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    srcs = [f for pat in ("*.hip", "*.h", "*.inc") for f in glob.glob(os.path.join(root, "parler_tts_amd", "csrc", pat))]
    read, dev = set(), set()
    for f in srcs:
        text = open(f).read()
        read |= set(re.findall(r'(?<![a-z_])getenv\("(PTTS_[A-Z0-9_]+)"\)', text))
        dev |= set(re.findall(r'ptts_dev_env\("(PTTS_[A-Z0-9_]+)"\)', text))
    assert 0 < len(read) <= 15, sorted(read)
    assert not (read & dev), sorted(read & dev)
    design = open(os.path.join(root, "DESIGN.md")).read()
    rows = set(re.findall(r"^\| `(PTTS_[A-Z0-9_]+)` \|", design, flags=re.M))
    assert rows == read, sorted(rows ^ read)
    tests_text = "".join(open(f).read() for f in glob.glob(os.path.join(root, "tests", "test_*gpu*.py")))
    for name in read:
        assert f'"{name}"' in tests_text, f"{name}: no GPU test sets it"
    from parler_tts_amd import _native as N

    blob = open(N.LIB_PATH, "rb").read()
    for name in read:
        assert name.encode() in blob, f"{name} missing from the built library (stale build?)"
    leaked = sorted(n for n in dev if n.encode() + b"\0" in blob)
    assert not leaked, f"development knobs compiled into the product library: {leaked}"
