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
from .distributed import broadcast_model_weights, generate_sharded, shard_batch, shard_range


def register_with_transformers() -> bool:
    """The reference's codec plug point (parler_tts/__init__.py:20-25): ``AutoConfig.register("dac_on_the_hub", DACConfig)`` and
    ``AutoModel.register(DACConfig, DACModel)``, so ``AutoConfig / AutoModel.from_pretrained(<dac dir>)`` resolve to this package's
    codec. transformers requires a ``PretrainedConfig`` subclass there: ``DACHubConfig`` carries exactly DACConfig's fields (plus
    any extra keys of the checkpoint's config.json) and ``DACModel`` accepts either. Returns False when transformers refuses
    (e.g. another package already registered the model type)."""
    try:
        from transformers import AutoConfig, AutoModel, PretrainedConfig
    except Exception:  # noqa: BLE001
        return False

    class DACHubConfig(PretrainedConfig):
        model_type = "dac_on_the_hub"

        def __init__(self, num_codebooks: int = 9, model_bitrate: int = 8, codebook_size: int = 1024, latent_dim: int = 1024,
                     frame_rate: int = 86, sampling_rate: int = 44100, **kwargs):
            self.num_codebooks, self.model_bitrate, self.codebook_size = num_codebooks, model_bitrate, codebook_size
            self.latent_dim, self.frame_rate, self.sampling_rate = latent_dim, frame_rate, sampling_rate
            super().__init__(**kwargs)

    try:
        AutoConfig.register("dac_on_the_hub", DACHubConfig, exist_ok=True)
        DACModel.config_class = DACHubConfig
        AutoModel.register(DACHubConfig, DACModel, exist_ok=True)
    except Exception:  # noqa: BLE001
        DACModel.config_class = DACConfig
        return False
    globals()["DACHubConfig"] = DACHubConfig
    return True


REGISTERED_WITH_TRANSFORMERS = register_with_transformers()

__all__ = ["DACConfig", "DACModel", "ParlerTTSConfig", "ParlerTTSDecoderConfig", "ParlerTTSForCausalLM",
           "ParlerTTSForConditionalGeneration", "ParlerTTSLogitsProcessor", "ParlerTTSStreamer",
           "apply_delay_pattern_mask", "build_delay_pattern_mask", "broadcast_model_weights", "generate_sharded", "shard_batch", "shard_range"]
