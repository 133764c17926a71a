"""CPU: the Python glue of ``generate()`` (conditioning, lengths, delay pattern, device-loop / host-loop drivers, streamer
hand-off, voice-prompt prefix, un-delay, special-id filtering, zero padding, return types) exercised WITHOUT a GPU: the two
native engines are replaced - in this test file only - by stand-ins that implement the engine interface with the oracle
(tests may use the oracle; the product never does). What the HIP engines compute is covered by the ``-m gpu`` tests; this
keeps the host logic around them under test in the CPU tier."""
import threading
import types

import pytest
import torch

from oracle import dac_oracle as DA
from oracle import decoder_oracle as DO

import parler_tts_amd as P


class OracleEngine:
    """Engine-interface stand-in: same methods and state semantics as parler_tts_amd.engine.DecoderEngine."""

    def __init__(self, spec, sd):
        self.spec, self.sd = spec, sd
        self.cfg = types.SimpleNamespace(max_batch=64, max_enc=4096, max_prompt=4096, max_ctx=1 << 20)
        self.prefix = None

    def set_gen_params(self, **kw):
        self.gp = DO.GenParams(max_length=kw["max_length"], min_new_tokens=kw.get("min_new_tokens", 0), do_sample=kw.get("do_sample", False),
                               temperature=kw.get("temperature", 1.0), top_k=kw.get("top_k", 0), top_p=kw.get("top_p", 1.0),
                               use_eos_gate=kw.get("use_eos_gate", True))
        self.max_length = kw["max_length"]

    def set_audio_prefix(self, codes):
        self.prefix = None if codes is None else codes.reshape(-1, codes.shape[-1]).cpu()

    def prefill(self, enc, enc_mask, prompt, prompt_mask, sample=True):
        self.args = (enc.float().cpu(), enc_mask.cpu() if enc_mask is not None else None, prompt.float().cpu() if prompt is not None else None,
                     prompt_mask.cpu() if prompt_mask is not None else None)
        self.orc = DO.DecoderOracle(self.spec, self.sd)
        K, bsz = self.spec.num_codebooks, enc.shape[0]
        seq = torch.full((bsz * K, 1), self.spec.bos_token_id, dtype=torch.long)
        if self.prefix is not None:
            seq = torch.cat([seq, self.prefix], dim=-1)
        self.seq, self.pattern = DO.build_delay_pattern_mask(seq, self.spec.bos_token_id, self.spec.pad_token_id, self.gp.max_length, K)
        self.prefix = None
        if sample:  # the device loop: run the oracle's whole loop once, hand columns out as decode_steps asks for them
            self.full = DO.sample_loop(self.orc, *self.args, self.gp, decoder_input_ids=self.seq[:, 1:] if self.seq.shape[1] > 1 else None).sequences
            if self.seq.shape[1] > 1:  # sample_loop re-applies the delay to un-delayed codes; here the prefix is already delayed: redo exactly
                self.full = self._loop_from_delayed()
            self.cur = self.seq.shape[1] + 1
        else:
            fed = DO.apply_delay_pattern_mask(self.seq, self.pattern)
            self._logits = self.orc.forward(fed, *self.args)[:, -1]

    def _loop_from_delayed(self):
        # greedy continuation from already-delayed given columns (what the HIP engine's teacher forcing + tail does)
        orc = DO.DecoderOracle(self.spec, self.sd)
        seq, K = self.seq.clone(), self.spec.num_codebooks
        gate = DO.EosGate(self.spec.eos_token_id, K, seq.shape[0] // K)
        unf = torch.ones(seq.shape[0], dtype=torch.long)
        given, first = seq.shape[1], True
        while True:
            fed = DO.apply_delay_pattern_mask(seq, self.pattern)
            lg = (orc.forward(fed, *self.args) if first else orc.forward(fed[:, -1:]))[:, -1].clone()
            first = False
            if self.gp.min_new_tokens > 0 and seq.shape[1] - given < self.gp.min_new_tokens:
                lg[:, self.spec.eos_token_id] = -float("inf")
            lg = gate(seq, lg)
            nxt = lg.argmax(-1) * unf + self.spec.pad_token_id * (1 - unf)
            seq = torch.cat([seq, nxt[:, None]], 1)
            unf = unf & ~((nxt == self.spec.eos_token_id) | (seq.shape[1] >= self.gp.max_length)).long()
            if unf.max() == 0:
                return seq

    def decode_steps(self, n):
        self.cur = min(self.cur + n, self.full.shape[1])

    def state(self):
        return self.cur, self.cur >= self.full.shape[1]

    def ids(self):
        return self.full[:, : self.cur]

    def logits(self):
        return self._logits

    def push_tokens(self, tokens, finished=None):
        self.seq = torch.cat([self.seq, tokens.cpu()[:, None]], dim=1)

    def step_forward(self):
        fed = DO.apply_delay_pattern_mask(self.seq, self.pattern)
        self._logits = self.orc.forward(fed[:, -1:])[:, -1]

    def close(self):
        pass


def _model(eos_gain=None):
    from transformers import T5Config

    torch.manual_seed(0)
    t5 = T5Config(vocab_size=128, d_model=128, d_kv=32, d_ff=256, num_layers=2, num_heads=4, feed_forward_proj="gated-gelu")
    dec = P.ParlerTTSDecoderConfig(vocab_size=1088, max_position_embeddings=256, num_hidden_layers=2, ffn_dim=256, num_attention_heads=2,
                                   hidden_size=128, num_codebooks=9, pad_token_id=1024, eos_token_id=1024, bos_token_id=1025)
    m = P.ParlerTTSForConditionalGeneration(P.ParlerTTSConfig.from_sub_models_config(t5, P.DACConfig(latent_dim=64, decoder_dim=256, decoder_rates=[4, 2, 2, 2]),
                                                                                    dec, vocab_size=128))
    spec, sd = DO.TINY, DO.make_decoder_weights(DO.TINY, seed=1237)
    for k in range(9):
        if eos_gain:
            sd[f"lm_heads.{k}.weight"][1024] *= eos_gain
        else:
            sd[f"lm_heads.{k}.weight"][1024:] = 0.0
    m.decoder.load_state_dict(sd, strict=False)
    dsd = DA.make_dac_weights(DA.DAC_TINY, seed=4321)
    eng = OracleEngine(spec, sd)
    m._get_engine = lambda B, N, Pp, L, T=0: eng
    dac = DA.DacOracle(DA.DAC_TINY, dsd)
    m.audio_encoder.decode = lambda audio_codes, audio_scales=None, **kw: types.SimpleNamespace(audio_values=dac.decode(audio_codes[0].cpu()))

    def decode_chunk(audio_codes, first_frame, n_frames=None, halo=16):  # ptts_dac_decode_chunk semantics on the oracle codec
        codes = audio_codes[0].cpu()
        n_frames = codes.shape[-1] - first_frame if n_frames is None else n_frames
        w0 = max(0, first_frame - halo)
        wav = dac.decode(codes[:, :, w0: first_frame + n_frames])
        hop = DA.DAC_TINY.hop_length
        return types.SimpleNamespace(audio_values=wav[:, :, (first_frame - w0) * hop:])

    m.audio_encoder.decode_chunk = decode_chunk

    def decode_filtered(audio_codes):  # ptts_dac_compact_codes + ptts_dac_decode_ragged semantics on the oracle codec
        codes = audio_codes[0].cpu()
        B, _, T = codes.shape
        hop = DA.DAC_TINY.hop_length
        out, frames = torch.zeros(B, 1, hop * T), torch.zeros(B, dtype=torch.int32)
        for b in range(B):
            ok = ((codes[b] >= 1024) | (codes[b] < 0)).sum(dim=0) == 0
            n = int(ok.sum())
            frames[b] = n
            if n:
                out[b, 0, :hop * n] = dac.decode(codes[b:b + 1, :, ok])[0, 0]
        return out, frames

    m.audio_encoder.decode_filtered = decode_filtered
    return m, spec, sd, dac


def _reference_pipeline(m, spec, sd, dac, desc, prompt_ids, gp, prefix=None):
    enc = m._encode_description(desc, None).float()
    prompt = m.embed_prompts(prompt_ids).float()
    tr = DO.sample_loop(DO.DecoderOracle(spec, sd), enc, None, prompt, None, gp, decoder_input_ids=prefix)
    codes = DO.undelay(tr.sequences, spec, gp.max_length, decoder_input_ids=prefix)
    return [dac.decode(DO.valid_frames(codes[b])[None])[0, 0] if DO.valid_frames(codes[b]).shape[1] else torch.zeros(1) for b in range(codes.shape[0])]


def test_device_loop_glue_with_eos_padding_and_return_dict():
    m, spec, sd, dac = _model(eos_gain=6.0)
    g = torch.Generator().manual_seed(1)
    desc, prompt_ids = torch.randint(3, 128, (2, 9), generator=g), torch.randint(3, 128, (2, 4), generator=g)
    out = m.generate(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_length=40, min_new_tokens=3, return_dict_in_generate=True)
    ref = _reference_pipeline(m, spec, sd, dac, desc, prompt_ids, DO.GenParams(max_length=40, min_new_tokens=3))
    assert out["audios_length"] == [int(w.shape[0]) for w in ref]
    for b, w in enumerate(ref):
        assert torch.allclose(out.sequences[b, : w.shape[0]], w, atol=1e-6) and float(out.sequences[b, w.shape[0]:].abs().max() if w.shape[0] < out.sequences.shape[1] else 0) == 0.0


def test_host_loop_glue_equals_device_loop_and_streamer_protocol():
    from transformers import LogitsProcessorList

    m, spec, sd, dac = _model()
    g = torch.Generator().manual_seed(2)
    desc, prompt_ids = torch.randint(3, 128, (1, 7), generator=g), torch.randint(3, 128, (1, 5), generator=g)
    kw = dict(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_new_tokens=24, min_new_tokens=24)
    a = m.generate(**kw)
    b = m.generate(logits_processor=LogitsProcessorList([P.ParlerTTSLogitsProcessor(1024, 9, 1, "cpu")]), **kw)  # manual (host-loop) path
    assert a.shape == (1, 32 * (25 - 9)) and torch.allclose(a, b, atol=1e-6)
    import numpy as np

    by_mode = []
    for inc in (True, False):  # generate() runs in a background thread, the caller iterates chunks (INFERENCE.md:141-148)
        streamer = P.ParlerTTSStreamer(m, device="cpu", play_steps=10, stride=8, incremental=inc)
        streamer.halo_frames = 64  # >= the whole tiny utterance: the window logic is exercised, exactness does not depend on the halo
        th = threading.Thread(target=lambda: m.generate(streamer=streamer, **kw))
        th.start()
        by_mode.append([c for c in streamer])
        th.join()
    chunks = by_mode[0]
    total = np.concatenate(chunks)
    n = len(chunks[-1])
    assert len(chunks) >= 3 and total.shape[0] == a.shape[1] and n > 0 and np.allclose(total[-n:], a[0].numpy()[-n:], atol=1e-5)
    assert [len(c) for c in by_mode[0]] == [len(c) for c in by_mode[1]] and all(np.allclose(x, y, atol=1e-6) for x, y in zip(*by_mode))


def test_voice_prompt_glue_prefix_survives_and_lengths_count_from_given_columns():
    m, spec, sd, dac = _model()
    g = torch.Generator().manual_seed(3)
    desc, prompt_ids = torch.randint(3, 128, (1, 7), generator=g), torch.randint(3, 128, (1, 5), generator=g)
    prefix = torch.randint(0, 1024, (9, 6), generator=g)
    kw = dict(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_new_tokens=20, min_new_tokens=20)
    a = m.generate(decoder_input_ids=prefix, **kw)
    ref = _reference_pipeline(m, spec, sd, dac, desc, prompt_ids, DO.GenParams(max_length=27, min_new_tokens=20), prefix=prefix)
    assert a.shape == (1, 32 * 18) and torch.allclose(a[0], ref[0], atol=1e-6)
    b = m.generate(decoder_input_ids=torch.cat([torch.full((9, 1), 1025), prefix], 1), **kw)  # leading BOS column is recognised (:3017-3018)
    assert torch.equal(a, b)
    with pytest.raises(ValueError, match="no room"):
        m.generate(decoder_input_ids=prefix, input_ids=desc, prompt_input_ids=prompt_ids, max_length=7)


def test_masks_encoder_outputs_and_num_return_sequences_glue():
    """Padded description / prompt masks travel to the engine untouched, `encoder_outputs` bypasses the text encoder, and
    `num_return_sequences` repeats every utterance (repeat_interleave, :3556-3561) - all host logic."""
    m, spec, sd, dac = _model()
    g = torch.Generator().manual_seed(5)
    desc, prompt_ids = torch.randint(3, 128, (2, 9), generator=g), torch.randint(3, 128, (2, 6), generator=g)
    dmask = torch.ones(2, 9, dtype=torch.long); dmask[1, 6:] = 0
    pmask = torch.ones(2, 6, dtype=torch.long); pmask[0, :2] = 0
    kw = dict(prompt_input_ids=prompt_ids, prompt_attention_mask=pmask, do_sample=False, max_new_tokens=24, min_new_tokens=24)
    a = m.generate(input_ids=desc, attention_mask=dmask, **kw)
    enc = m._encode_description(desc, dmask)
    b = m.generate(encoder_outputs=(enc,), attention_mask=dmask, **kw)  # tuple form of BaseModelOutput
    assert a.shape == (2, 32 * 16) and torch.equal(a, b)
    # oracle pipeline with the same masks
    prompt = m.embed_prompts(prompt_ids).float()
    gp = DO.GenParams(max_length=25, min_new_tokens=24)
    tr = DO.sample_loop(DO.DecoderOracle(spec, sd), enc.float(), dmask, prompt, pmask, gp)
    codes = DO.undelay(tr.sequences, spec, 25)
    for bb in range(2):
        assert torch.allclose(a[bb], dac.decode(DO.valid_frames(codes[bb])[None])[0, 0], atol=1e-4)
    # num_return_sequences: every utterance repeated in place (greedy needs sampling mode in GenerationConfig: top_k=1 keeps it deterministic
    # on the manual path, where torch.multinomial over a one-hot distribution has a single outcome)
    from transformers import LogitsProcessorList

    c = m.generate(input_ids=desc, attention_mask=dmask, num_return_sequences=2, do_sample=True, top_k=1,
                   logits_processor=LogitsProcessorList([P.ParlerTTSLogitsProcessor(1024, 9, 4, "cpu")]),
                   prompt_input_ids=prompt_ids, prompt_attention_mask=pmask, max_new_tokens=24, min_new_tokens=24)
    assert c.shape == (4, 32 * 16)
    assert torch.allclose(c[0], a[0], atol=1e-4) and torch.allclose(c[1], a[0], atol=1e-4) and torch.allclose(c[2], a[1], atol=1e-4) and torch.allclose(c[3], a[1], atol=1e-4)


def test_device_loop_polls_the_device_only_once_eos_can_fire():
    """The default loop synchronises with the device (ptts_state) only where a row could have finished: never while `min_new_tokens`
    still blocks EOS, once per 64-step chunk afterwards, and once at max_length. Same ids either way."""
    m, spec, sd, dac = _model()
    eng = m._get_engine(1, 0, 0, 0)
    polls = []
    orig_state = eng.state
    eng.state = lambda: (polls.append(1), orig_state())[1]
    g = torch.Generator().manual_seed(3)
    desc, prompt_ids = torch.randint(3, 128, (1, 9), generator=g), torch.randint(3, 128, (1, 4), generator=g)
    a = m.generate(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_new_tokens=150, min_new_tokens=150)
    assert len(polls) == 1  # EOS blocked throughout: one poll, at max_length
    polls.clear()
    b = m.generate(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_new_tokens=150, min_new_tokens=70)
    assert len(polls) == 2  # chunks end at 65, 129, 150 generated columns: 65 <= 70 is not polled, 129 and 150 are
    polls.clear()
    c = m.generate(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_new_tokens=150, min_new_tokens=0)
    assert len(polls) == 3
    assert torch.equal(a, b) and torch.equal(a, c)  # EOS never wins in this model (its logit rows are zeroed): the same utterance
