"""Generation options of ``GenerationConfig`` beyond what the device-resident sampler implements.

The reference's ``generate()`` hands its ``GenerationConfig`` to transformers' ``_get_logits_processor`` /
``_get_stopping_criteria`` (parler_tts/modeling_parler_tts.py:3540-3552), so EVERY processor transformers derives from the
config is honoured there: repetition / n-gram penalties, bad words, suppressed / forced tokens, min-p, typical-p, epsilon /
eta cut-offs, logit renormalisation, ``max_time`` ... The HIP sampler (``tail_kernel``) implements the ones Parler-TTS is
run with - min_new_tokens / min_length, the EOS gate, temperature, top-k, top-p - on the device. When a call asks for
anything else, ``generate()`` switches to its host loop (HIP forward, selection in torch) and applies the processors built
here: transformers' OWN classes, instantiated with the arguments and in the order of its ``_get_logits_processor``
(config processors -> the caller's / default ``ParlerTTSLogitsProcessor`` list -> warpers when sampling ->
``LogitNormalization``). Nothing is silently ignored: options that cannot be honoured raise.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def _get(gc, name, default=None):
    return getattr(gc, name, default)


def check_generation_mode(gc) -> None:
    """Greedy or multinomial sampling only (:3574-3578): beam search, assisted / contrastive / DoLa decoding are refused with the
    reference's message."""
    msg = ("Got incompatible mode for generation, should be one of greedy or sampling. "
           "Ensure that beam search is de-activated by setting `num_beams=1` and `num_beam_groups=1`.")
    if (_get(gc, "num_beams", 1) or 1) > 1 or (_get(gc, "num_beam_groups", 1) or 1) > 1:
        raise ValueError(msg)
    if _get(gc, "penalty_alpha") or _get(gc, "dola_layers") is not None or _get(gc, "prompt_lookup_num_tokens") is not None \
            or _get(gc, "assistant_early_exit") is not None:
        raise ValueError(msg)


# keyword arguments of the reference's forward / prepare_inputs_for_generation (:2695-2720, :2882-2899) that reach `generate()` as model
# kwargs. The first group has no effect on what this path computes (or is consumed elsewhere); the second changes the computation in the
# reference and is not implemented here: refused when given, never silently dropped.
_NOOP_MODEL_KWARGS = {"use_cache", "return_dict", "output_attentions", "output_hidden_states", "padding_mask", "past_key_values", "cache_position",
                      "labels", "loss_reduction", "tokenizer", "assistant_tokenizer"}
_UNSUPPORTED_MODEL_KWARGS = {"inputs_embeds", "decoder_inputs_embeds", "decoder_attention_mask", "decoder_position_ids", "head_mask",
                             "decoder_head_mask", "cross_attn_head_mask", "negative_prompt_ids", "negative_prompt_attention_mask",
                             "assistant_model", "prefix_allowed_tokens_fn"}


def check_model_kwargs(mk: dict) -> None:
    """``mk``: what is left of ``generate()``'s kwargs after the GenerationConfig fields and the arguments this path consumes."""
    bad = [k for k, v in mk.items() if k in _UNSUPPORTED_MODEL_KWARGS and v is not None]
    if bad:
        raise NotImplementedError(f"generate() arguments {bad} change the reference's computation and are not implemented by the HIP generation path")
    unused = [k for k, v in mk.items() if k not in _NOOP_MODEL_KWARGS and k not in _UNSUPPORTED_MODEL_KWARGS and v is not None]
    if unused:
        raise ValueError(f"The following `model_kwargs` are not used by the model: {unused} (note: typos in the generate arguments will also show up "
                         "in this list)")


def active_extras(gc) -> List[str]:
    """Names of the config options in effect that the device sampler does not implement (empty for the usual Parler-TTS calls)."""
    on: List[str] = []

    def ne(name, neutral):
        v = _get(gc, name)
        if v is not None and v != neutral:
            on.append(name)

    ne("repetition_penalty", 1.0)
    ne("encoder_repetition_penalty", 1.0)
    ne("no_repeat_ngram_size", 0)
    ne("encoder_no_repeat_ngram_size", 0)
    for name in ("bad_words_ids", "sequence_bias", "forced_bos_token_id", "forced_eos_token_id", "exponential_decay_length_penalty",
                 "suppress_tokens", "begin_suppress_tokens", "max_time", "stop_strings", "watermarking_config"):
        if _get(gc, name) is not None:
            on.append(name)
    if _get(gc, "remove_invalid_values") is True:
        on.append("remove_invalid_values")
    if _get(gc, "renormalize_logits") is True:
        on.append("renormalize_logits")
    ne("guidance_scale", 1)
    if _get(gc, "do_sample"):
        for name in ("min_p", "top_h"):
            if _get(gc, name) is not None:
                on.append(name)
        v = _get(gc, "typical_p")
        if v is not None and v < 1.0:
            on.append("typical_p")
        for name in ("epsilon_cutoff", "eta_cutoff"):
            v = _get(gc, name)
            if v is not None and 0.0 < v < 1.0:
                on.append(name)
    return on


def build_processors(gc, input_ids_seq_length: int, encoder_input_ids: Optional[torch.Tensor], custom, device, eos_token_id: int):
    """The complete ``LogitsProcessorList`` of one call, as ``GenerationMixin._get_logits_processor`` assembles it, from transformers'
    own processor classes. ``custom``: the caller's list, or the default ``[ParlerTTSLogitsProcessor]`` (:3418)."""
    from transformers.generation import logits_process as LP

    if _get(gc, "guidance_scale") is not None and gc.guidance_scale != 1:
        raise NotImplementedError("guidance_scale != 1 (classifier-free guidance needs a second, unconditional decoder pass per step) is not "
                                  "implemented by the HIP generation path")
    if _get(gc, "watermarking_config") is not None:
        raise NotImplementedError("watermarking_config is not supported by the HIP generation path")
    eos = torch.tensor([eos_token_id], device=device)
    procs = LP.LogitsProcessorList()
    if _get(gc, "sequence_bias") is not None:
        procs.append(LP.SequenceBiasLogitsProcessor(sequence_bias=gc.sequence_bias))
    enc2d = encoder_input_ids is not None and encoder_input_ids.dim() == 2
    v = _get(gc, "encoder_repetition_penalty")
    if v is not None and v != 1.0 and enc2d:
        procs.append(LP.EncoderRepetitionPenaltyLogitsProcessor(penalty=v, encoder_input_ids=encoder_input_ids))
    v = _get(gc, "repetition_penalty")
    if v is not None and v != 1.0:
        procs.append(LP.RepetitionPenaltyLogitsProcessor(penalty=v))
    v = _get(gc, "no_repeat_ngram_size")
    if v is not None and v > 0:
        procs.append(LP.NoRepeatNGramLogitsProcessor(v))
    v = _get(gc, "encoder_no_repeat_ngram_size")
    if v is not None and v > 0 and enc2d:
        procs.append(LP.EncoderNoRepeatNGramLogitsProcessor(v, encoder_input_ids))
    if _get(gc, "bad_words_ids") is not None:
        procs.append(LP.NoBadWordsLogitsProcessor(gc.bad_words_ids, eos))
    v = _get(gc, "min_length")
    if v is not None and v > 0:
        procs.append(LP.MinLengthLogitsProcessor(v, eos, device=device))
    v = _get(gc, "min_new_tokens")
    if v is not None and v > 0:
        procs.append(LP.MinNewTokensLengthLogitsProcessor(input_ids_seq_length, v, eos, device=device))
    if _get(gc, "forced_bos_token_id") is not None:
        procs.append(LP.ForcedBOSTokenLogitsProcessor(gc.forced_bos_token_id))
    if _get(gc, "forced_eos_token_id") is not None:
        procs.append(LP.ForcedEOSTokenLogitsProcessor(gc.max_length, gc.forced_eos_token_id, device=device))
    if _get(gc, "remove_invalid_values") is True:
        procs.append(LP.InfNanRemoveLogitsProcessor())
    if _get(gc, "exponential_decay_length_penalty") is not None:
        procs.append(LP.ExponentialDecayLengthPenalty(gc.exponential_decay_length_penalty, eos, input_ids_seq_length))
    if _get(gc, "suppress_tokens") is not None:
        procs.append(LP.SuppressTokensLogitsProcessor(gc.suppress_tokens, device=device))
    if _get(gc, "begin_suppress_tokens") is not None:
        begin = input_ids_seq_length if (input_ids_seq_length > 1 or _get(gc, "forced_bos_token_id") is None) else input_ids_seq_length + 1
        procs.append(LP.SuppressTokensAtBeginLogitsProcessor(gc.begin_suppress_tokens, begin, device=device))
    procs.extend(custom or [])  # _merge_criteria_processor_list: config processors first, then the custom list
    if _get(gc, "do_sample"):
        v = _get(gc, "temperature")
        if v is not None and v != 1.0:
            procs.append(LP.TemperatureLogitsWarper(v))
        if _get(gc, "top_h") is not None:
            if not hasattr(LP, "TopHLogitsWarper"):
                raise NotImplementedError("top_h needs a transformers release that ships TopHLogitsWarper")
            procs.append(LP.TopHLogitsWarper(top_h=gc.top_h))
        v = _get(gc, "top_k")
        if v is not None and v != 0:
            procs.append(LP.TopKLogitsWarper(top_k=v, min_tokens_to_keep=1))
        v = _get(gc, "top_p")
        if v is not None and v < 1.0:
            procs.append(LP.TopPLogitsWarper(top_p=v, min_tokens_to_keep=1))
        if _get(gc, "min_p") is not None:
            procs.append(LP.MinPLogitsWarper(min_p=gc.min_p, min_tokens_to_keep=1))
        v = _get(gc, "typical_p")
        if v is not None and v < 1.0:
            procs.append(LP.TypicalLogitsWarper(mass=v, min_tokens_to_keep=1))
        v = _get(gc, "epsilon_cutoff")
        if v is not None and 0.0 < v < 1.0:
            procs.append(LP.EpsilonLogitsWarper(epsilon=v, min_tokens_to_keep=1))
        v = _get(gc, "eta_cutoff")
        if v is not None and 0.0 < v < 1.0:
            procs.append(LP.EtaLogitsWarper(epsilon=v, min_tokens_to_keep=1, device=device))
    if _get(gc, "renormalize_logits") is True:
        procs.append(LP.LogitNormalization())
    return procs


def build_stopping_criteria(gc, user_criteria) -> Tuple[list, list]:
    """(criteria from the config, the caller's criteria). max_length and EOS are handled by the loop itself; ``max_time`` becomes
    transformers' ``MaxTimeCriteria``; ``stop_strings`` needs the tokenizer ``generate()`` of the reference never forwards (:3550), so it
    raises like transformers does."""
    from transformers.generation import stopping_criteria as SC

    if _get(gc, "stop_strings") is not None:
        raise ValueError("There are one or more stop strings, either in the arguments to `generate` or in the model's generation config, but "
                         "we could not locate a tokenizer. When generating with stop strings, you must pass the model's tokenizer to the "
                         "`tokenizer` argument of `generate`.")
    cfg = []
    if _get(gc, "max_time") is not None:
        cfg.append(SC.MaxTimeCriteria(max_time=gc.max_time))
    return cfg, list(user_criteria or [])
