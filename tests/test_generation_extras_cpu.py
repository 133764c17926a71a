"""CPU: GenerationConfig options beyond the device sampler (parler_tts_amd/generation_extras.py). The reference forwards its config to
transformers' `_get_logits_processor` (modeling_parler_tts.py:3540-3547), so repetition / n-gram penalties, bad words, min-p, typical-p,
renormalisation ... are honoured there; here they run the host loop with transformers' own processor objects. Pinned against the
INSTALLED transformers: (a) the processor list equals what its `_get_logits_processor` builds (classes, order, effect on scores),
(b) `generate()` (host glue + oracle-backed engine stand-ins, as in test_generate_glue_cpu.py) produces the audio of the ids its `_sample`
produces with the same config. The default path is untouched: no extra option -> device loop."""
import copy

import pytest
import torch
from transformers import GenerationConfig

from oracle import decoder_oracle as DO
from oracle import hf_sample_shim as HS

import parler_tts_amd as P
from parler_tts_amd import generation_extras as GX

from test_generate_glue_cpu import _model  # noqa: E402 — the oracle-backed stand-ins of the generate() glue tests


def _gate(spec, bsz):
    return P.ParlerTTSLogitsProcessor(spec.eos_token_id, spec.num_codebooks, bsz, "cpu")


EXTRA_CASES = [
    dict(do_sample=False, repetition_penalty=1.3, no_repeat_ngram_size=3),
    dict(do_sample=False, bad_words_ids=[[5], [7, 9]], suppress_tokens=[3, 11], begin_suppress_tokens=[2], min_length=6),
    dict(do_sample=False, forced_eos_token_id=1024, exponential_decay_length_penalty=(4, 1.05), remove_invalid_values=True, renormalize_logits=True),
    dict(do_sample=True, temperature=0.8, top_k=40, top_p=0.95, min_p=0.02, typical_p=0.9, epsilon_cutoff=3e-4, eta_cutoff=2e-4, repetition_penalty=1.1),
    dict(do_sample=True, encoder_repetition_penalty=1.2, encoder_no_repeat_ngram_size=2, sequence_bias=[[[5], -2.0], [[7, 9], 1.5]]),
]


@pytest.mark.parametrize("extra", EXTRA_CASES)
def test_processor_list_equals_what_transformers_builds(extra):
    spec = DO.TINY
    bsz, given = 2, 1
    rows = bsz * spec.num_codebooks
    gc = GenerationConfig(max_length=30, min_new_tokens=4, pad_token_id=spec.pad_token_id, eos_token_id=spec.eos_token_id,
                          bos_token_id=spec.bos_token_id, **extra)
    enc_ids = torch.randint(3, 100, (rows, 6), generator=torch.Generator().manual_seed(3))  # rows must match the scores for the encoder-* processors
    mine = GX.build_processors(copy.deepcopy(gc), given, enc_ids, [_gate(spec, bsz)], torch.device("cpu"), spec.eos_token_id)
    host = HS.OracleGenerationHost(DO.DecoderOracle(spec, DO.make_decoder_weights(spec, seed=1)), None, None, None, None)
    gch = copy.deepcopy(gc)
    host._prepare_special_tokens(gch, False, device=torch.device("cpu"))
    from transformers.generation.logits_process import LogitsProcessorList

    theirs = host._get_logits_processor(generation_config=gch, input_ids_seq_length=given, encoder_input_ids=enc_ids,
                                        prefix_allowed_tokens_fn=None, logits_processor=LogitsProcessorList([_gate(spec, bsz)]), device="cpu")
    assert [type(p).__name__ for p in mine] == [type(p).__name__ for p in theirs]
    assert set(GX.active_extras(gc)) >= {k for k in extra if k not in ("do_sample", "temperature", "top_k", "top_p", "min_length")}
    g = torch.Generator().manual_seed(11)
    for t in (1, 2, 5, 9):  # the same scores through both lists at several sequence lengths (stateful EOS gate: fresh lists per side)
        ids = torch.randint(0, 1024, (rows, t), generator=g)
        ids[:, 0] = spec.bos_token_id
        scores = torch.randn(rows, spec.vocab_size, generator=g) * 3
        a, b = mine(ids, scores.clone()), theirs(ids, scores.clone())
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(a[~torch.isnan(a)], b[~torch.isnan(b)]), (t, extra)


@pytest.mark.parametrize("extra", [EXTRA_CASES[0], EXTRA_CASES[1], EXTRA_CASES[3]])
def test_generate_with_config_processors_equals_transformers_sample(extra):
    m, spec, sd, dac = _model(eos_gain=4.0)
    g = torch.Generator().manual_seed(21)
    desc, prompt_ids = torch.randint(3, 128, (2, 8), generator=g), torch.randint(3, 128, (2, 4), generator=g)
    kw = dict(input_ids=desc, prompt_input_ids=prompt_ids, max_length=28, min_new_tokens=4, return_dict_in_generate=True, **extra)
    torch.manual_seed(5)
    out = m.generate(**kw)
    # transformers' own `_sample` + `_get_logits_processor` on the same model (the oracle), config and RNG stream
    enc = m._encode_description(desc, None).float()
    prompt = m.embed_prompts(prompt_ids).float()
    gp = DO.GenParams(max_length=28, min_new_tokens=4, do_sample=extra["do_sample"], temperature=extra.get("temperature", 1.0),
                      top_k=extra.get("top_k", 0), top_p=extra.get("top_p", 1.0))
    rest = {k: v for k, v in extra.items() if k not in ("do_sample", "temperature", "top_k", "top_p")}
    torch.manual_seed(5)
    if extra["do_sample"]:
        torch.randint(0, 2 ** 62, (1,))  # generate() draws the device sampler's seed first (unused on the host loop)
    with torch.no_grad():
        seq, _, names = HS.hf_sample(DO.DecoderOracle(spec, sd), enc, None, prompt, None, gp, lambda b: _gate(spec, b), **rest)
    assert any(n in names for n in ("RepetitionPenaltyLogitsProcessor", "NoBadWordsLogitsProcessor", "MinPLogitsWarper"))
    codes = DO.undelay(seq, spec, 28)
    want = [dac.decode(DO.valid_frames(codes[b])[None])[0, 0] if DO.valid_frames(codes[b]).shape[1] else torch.zeros(1) for b in range(2)]
    assert out["audios_length"] == [int(w.shape[0]) for w in want]
    for b, w in enumerate(want):
        assert torch.allclose(out.sequences[b, : w.shape[0]], w, atol=1e-6)
    if "bad_words_ids" in extra:  # banning a handful of the 1024 codes need not change a 27-column utterance
        return
    # and the option really changed the result: the same call without it gives different audio
    torch.manual_seed(5)
    base = m.generate(**{k: v for k, v in kw.items() if k in ("input_ids", "prompt_input_ids", "max_length", "min_new_tokens", "return_dict_in_generate",
                                                              "do_sample", "temperature", "top_k", "top_p")})
    assert base.sequences.shape != out.sequences.shape or not torch.allclose(base.sequences, out.sequences, atol=1e-6)


def test_default_calls_stay_on_the_device_loop(monkeypatch):
    m, spec, sd, dac = _model()
    called = []
    monkeypatch.setattr(type(m), "_run_host_loop", lambda self, *a, **k: called.append("host") or (_ for _ in ()).throw(AssertionError("host loop")))
    g = torch.Generator().manual_seed(2)
    desc, prompt_ids = torch.randint(3, 128, (1, 7), generator=g), torch.randint(3, 128, (1, 5), generator=g)
    m.generate(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_new_tokens=12, min_new_tokens=12)
    m.generate(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=True, temperature=0.7, top_k=20, top_p=0.9, max_new_tokens=12, min_new_tokens=12,
               repetition_penalty=1.0, no_repeat_ngram_size=0, use_cache=True)
    assert not called


def test_max_time_becomes_a_stopping_criterion_and_unsupported_options_raise():
    m, spec, sd, dac = _model()
    g = torch.Generator().manual_seed(2)
    desc, prompt_ids = torch.randint(3, 128, (1, 7), generator=g), torch.randint(3, 128, (1, 5), generator=g)
    kw = dict(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False)
    full = m.generate(max_new_tokens=30, min_new_tokens=30, **kw)
    cut = m.generate(max_new_tokens=30, min_new_tokens=30, max_time=0.0, **kw)  # MaxTimeCriteria fires after the first generated column
    assert cut.shape[1] < full.shape[1]
    with pytest.raises(NotImplementedError, match="guidance_scale"):
        m.generate(max_new_tokens=12, guidance_scale=3.0, **kw)
    with pytest.raises(ValueError, match="tokenizer"):
        m.generate(max_new_tokens=12, stop_strings=["a"], **kw)
    with pytest.raises(ValueError, match="greedy or sampling"):
        m.generate(max_new_tokens=12, penalty_alpha=0.6, top_k=4, **kw)
    with pytest.raises(ValueError, match="not used by the model"):
        m.generate(max_new_tokens=12, prompt_inptu_ids=prompt_ids, **kw)
    with pytest.raises(NotImplementedError, match="inputs_embeds"):
        m.generate(max_new_tokens=12, inputs_embeds=torch.zeros(1, 7, 128), **kw)
    m.generate(max_new_tokens=12, min_new_tokens=12, use_cache=True, padding_mask=None, **kw)  # accepted no-ops


def test_output_scores_and_logits_are_the_per_step_tuples_of_transformers_sample():
    """return_dict_in_generate + output_scores / output_logits: the reference hands back `_sample`'s ModelOutput with the waveform in
    `.sequences` (:3648-3651), so `.scores` (processed) and `.logits` (raw) are there; here they come from the host loop."""
    m, spec, sd, dac = _model(eos_gain=4.0)
    g = torch.Generator().manual_seed(23)
    desc, prompt_ids = torch.randint(3, 128, (2, 8), generator=g), torch.randint(3, 128, (2, 4), generator=g)
    out = m.generate(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_length=24, min_new_tokens=5, return_dict_in_generate=True,
                     output_scores=True, output_logits=True)
    enc = m._encode_description(desc, None).float()
    prompt = m.embed_prompts(prompt_ids).float()
    with torch.no_grad():
        seq, scores, _ = HS.hf_sample(DO.DecoderOracle(spec, sd), enc, None, prompt, None, DO.GenParams(max_length=24, min_new_tokens=5), lambda b: _gate(spec, b))
    assert len(out.scores) == len(scores) == len(out.logits) == seq.shape[1] - 1
    for a, b, raw in zip(out.scores, scores, out.logits):
        assert torch.equal(torch.isinf(a), torch.isinf(b)) and torch.allclose(a[~torch.isinf(a)], b[~torch.isinf(b)], atol=1e-6)
        assert not torch.isinf(raw).any() and torch.allclose(raw[~torch.isinf(a)], a[~torch.isinf(a)], atol=1e-6)  # greedy: processing only masks
    plain = m.generate(input_ids=desc, prompt_input_ids=prompt_ids, do_sample=False, max_length=24, min_new_tokens=5, return_dict_in_generate=True)
    assert "scores" not in plain and torch.equal(plain.sequences, out.sequences)
    with pytest.raises(NotImplementedError, match="output_attentions"):
        m.generate(input_ids=desc, prompt_input_ids=prompt_ids, max_length=24, return_dict_in_generate=True, output_attentions=True)
