"""TEST INFRASTRUCTURE ONLY — lets the INSTALLED transformers ``GenerationMixin._sample`` drive the decoder oracle.

The reference calls ``self._sample(delayed_input_ids, logits_processor=..., stopping_criteria=..., generation_config=...,
**model_kwargs)`` (modeling_parler_tts.py:3564) on transformers 4.46.1, which is not installable here; the composite
reference model cannot even be constructed on the installed 5.x. What CAN run is the installed
release's own loop: this shim is a minimal ``GenerationMixin`` host whose ``forward`` is ``DecoderOracle.forward`` and
whose ``prepare_inputs_for_generation`` restates the two things the reference's does for this path (:2909 apply the
delay-pattern mask to the whole sequence, :2930 feed only the new column once the cache exists). Everything else —
processor list construction and ORDER, MinNewTokens, warpers, arg-max / multinomial, finished-row padding, concatenation,
EOS / max-length stopping — is transformers' code, so ``tests/test_sample_loop_vs_transformers.py`` pins
``decoder_oracle.sample_loop`` against it.
"""
from __future__ import annotations

from contextlib import nullcontext
from types import SimpleNamespace
from typing import Optional

import torch
from transformers import GenerationConfig
from transformers.generation.logits_process import LogitsProcessorList
from transformers.generation.stopping_criteria import StoppingCriteriaList
from transformers.generation.utils import GenerationMixin

from . import decoder_oracle as DO


class OracleGenerationHost(torch.nn.Module, GenerationMixin):
    def __init__(self, oracle: DO.DecoderOracle, enc, enc_mask, prompt, prompt_mask):
        super().__init__()
        self.oracle = oracle
        self.cond = (enc, enc_mask, prompt, prompt_mask)
        self.config = SimpleNamespace(is_encoder_decoder=False, max_position_embeddings=oracle.spec.max_position_embeddings)
        self._p = torch.nn.Parameter(torch.zeros(1))  # gives the mixin a device

    @property
    def device(self):
        return self._p.device

    # -- hooks `_sample` / `_prefill` call -------------------------------------------------------------------------
    def prepare_inputs_for_generation(self, input_ids, next_sequence_length=None, is_first_iteration=False,
                                      decoder_delay_pattern_mask=None, **kwargs):
        fed = DO.apply_delay_pattern_mask(input_ids, decoder_delay_pattern_mask)  # :2909
        first = self.oracle.past_len == 0
        return {"input_ids": fed if first else fed[:, -1:], "first": first}  # :2930

    def forward(self, input_ids=None, first=False, return_dict=True, **kwargs):
        if first:
            enc, enc_mask, prompt, prompt_mask = self.cond
            logits = self.oracle.forward(input_ids, enc, enc_mask, prompt, prompt_mask)
        else:
            logits = self.oracle.forward(input_ids)
        return SimpleNamespace(logits=logits)

    def _update_model_kwargs_for_generation(self, outputs, model_kwargs, is_encoder_decoder=False, **kw):
        return model_kwargs  # the cache lives inside the oracle

    def _valid_auto_compile_criteria(self, model_kwargs, generation_config):
        return False

    def _optimize_model_for_decode(self):
        return nullcontext()


def hf_sample(oracle: DO.DecoderOracle, enc, enc_mask, prompt, prompt_mask, gp: DO.GenParams, eos_gate_factory=None,
              decoder_input_ids: Optional[torch.Tensor] = None, encoder_input_ids: Optional[torch.Tensor] = None, **config_extra):
    """Runs the installed transformers `_sample` the way the reference's generate() sets it up (:3395-3572).
    Returns (sequences [B*K, Lout], processed scores per step). ``config_extra``: further GenerationConfig fields (repetition_penalty,
    no_repeat_ngram_size, min_p, ...) that transformers' own `_get_logits_processor` turns into processors (:3540-3547);
    ``encoder_input_ids``: the description token ids the reference passes there as `encoder_input_ids=inputs_tensor`."""
    spec = oracle.spec
    K = spec.num_codebooks
    bsz = enc.shape[0]
    oracle.reset()
    host = OracleGenerationHost(oracle, enc, enc_mask, prompt, prompt_mask)
    gc = GenerationConfig(do_sample=gp.do_sample, max_length=gp.max_length, min_new_tokens=gp.min_new_tokens or None,
                          temperature=gp.temperature if gp.do_sample else None, top_k=(gp.top_k or None) if gp.do_sample else None,
                          top_p=gp.top_p if gp.do_sample else None, pad_token_id=spec.pad_token_id, eos_token_id=spec.eos_token_id,
                          bos_token_id=spec.bos_token_id, return_dict_in_generate=True, output_scores=True, use_cache=True, **config_extra)
    host._prepare_special_tokens(gc, False, device=torch.device("cpu"))
    seq = torch.full((bsz * K, 1), spec.bos_token_id, dtype=torch.long)  # :3011-3014
    if decoder_input_ids is not None and decoder_input_ids.shape[-1] > 0:
        seq = torch.cat([seq, decoder_input_ids.long()], dim=-1)
    input_ids_length = seq.shape[-1]
    custom = LogitsProcessorList([eos_gate_factory(bsz)] if (gp.use_eos_gate and eos_gate_factory is not None) else [])  # :3418
    processors = host._get_logits_processor(generation_config=gc, input_ids_seq_length=input_ids_length, encoder_input_ids=encoder_input_ids,
                                            prefix_allowed_tokens_fn=None, logits_processor=custom, device="cpu")  # :3412-3427
    criteria = host._get_stopping_criteria(generation_config=gc, stopping_criteria=StoppingCriteriaList())  # :3424-3427
    delayed, pattern = DO.build_delay_pattern_mask(seq, spec.bos_token_id, spec.pad_token_id, gp.max_length, K)  # :3523-3530
    out = host._sample(delayed, logits_processor=processors, stopping_criteria=criteria, generation_config=gc, synced_gpus=False,
                       streamer=None, decoder_delay_pattern_mask=pattern, use_cache=True)  # :3564
    return out.sequences, list(out.scores), [type(p).__name__ for p in processors]
