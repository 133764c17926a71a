"""``dac_wrapper`` interface of the reference (configuration + ``DACModel.decode`` / ``.encode``), backed by the HIP DAC engine."""
from .modeling_dac import DACDecoderOutput, DACEncoderOutput, DACModel  # noqa: F401
from ..configuration_parler_tts import DACConfig  # noqa: F401  (the config lives with the other configs)

__all__ = ["DACConfig", "DACModel", "DACDecoderOutput", "DACEncoderOutput"]
