"""parler_tts_amd — MI355X-native Parler-TTS generation path: a drop-in for ``parler_tts`` on
``ParlerTTSForConditionalGeneration.from_pretrained()/.generate()``, ``ParlerTTSStreamer`` and the ``dac_wrapper``
interface (reference parler_tts/__init__.py:6-25). The decoder step and DAC decode run in libptts_hip.so
(hand-written HIP for gfx950, C ABI in include/ptts.h); there is no CPU fallback on the product path."""
__version__ = "0.1.0"

from .configuration_parler_tts import DACConfig, ParlerTTSConfig, ParlerTTSDecoderConfig
from .dac_wrapper import DACModel
from .logits_processors import ParlerTTSLogitsProcessor
from .modeling_parler_tts import (
    ParlerTTSForCausalLM,
    ParlerTTSForConditionalGeneration,
    apply_delay_pattern_mask,
    build_delay_pattern_mask,
)
from .streamer import ParlerTTSStreamer

__all__ = ["DACConfig", "DACModel", "ParlerTTSConfig", "ParlerTTSDecoderConfig", "ParlerTTSForCausalLM",
           "ParlerTTSForConditionalGeneration", "ParlerTTSLogitsProcessor", "ParlerTTSStreamer",
           "apply_delay_pattern_mask", "build_delay_pattern_mask"]
