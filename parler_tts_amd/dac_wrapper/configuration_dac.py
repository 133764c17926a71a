from ..configuration_parler_tts import DACConfig  # noqa: F401  (reference: dac_wrapper/configuration_dac.py:7-27)
